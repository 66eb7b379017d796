"""tests/golden/config_values.json: the values of the reference's config.yaml that the product hard-codes as defaults (build container).

    python oracle/make_golden_config.py

Plain YAML read of /root/reference/code/config.yaml (no Lightning / jsonargparse): sampler, guider, discretization, network_config,
ControlNet, trainer / inference parameters, decoder.  tests/test_host_logic.py compares the product's defaults with the committed JSON.
"""
import json
import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with open("/root/reference/code/config.yaml") as f:
        cfg = yaml.safe_load(f)
    ia = cfg["model"]["init_args"]
    mc = ia["module_loader"]["init_args"]["module_config"]
    net = mc["network_config"]["init_args"]
    smp = mc["sampler"]["init_args"]
    dec = mc["first_stage_model"]["init_args"]["decoder_config"]["params"]
    out = dict(
        seed_everything=cfg["seed_everything"], precision=cfg["trainer"]["precision"],
        network={k: net[k] for k in ("in_channels", "model_channels", "out_channels", "num_res_blocks", "attention_resolutions", "channel_mult",
                                     "num_head_channels", "context_dim", "adm_in_channels", "controlnet_mode", "use_apm", "merging_mode",
                                     "extra_ff_mix_layer", "use_spatial_context", "merge_strategy", "merge_factor", "video_kernel_size",
                                     "use_linear_in_transformer", "max_ddpm_temb_period", "transformer_depth")},
        controlnet={k: mc["controlnet"]["init_args"]["model_params"][k] for k in ("conditioning_embedding_out_channels", "merging_mode", "zero_conv_mode",
                                                                                    "downsample_controlnet_cond", "use_image_encoder_normalization")},
        sampler=dict(num_steps=smp["num_steps"], s_churn=smp["s_churn"], s_tmin=smp["s_tmin"], s_noise=smp["s_noise"],
                     discretization=smp["discretization_config"]["target"].split(".")[-1], sigma_max=smp["discretization_config"]["params"]["sigma_max"],
                     guider=smp["guider_config"]["target"].split(".")[-1], **{k: smp["guider_config"]["params"][k] for k in ("max_scale", "min_scale", "num_frames")}),
        denoiser_scaling=mc["denoiser"]["init_args"]["scaling_config"]["target"].split(".")[-1],
        decoder={k: dec[k] for k in ("ch", "ch_mult", "num_res_blocks", "z_channels", "out_ch", "video_kernel_size")} if "video_kernel_size" in dec else
                {k: dec[k] for k in ("ch", "ch_mult", "num_res_blocks", "z_channels", "out_ch")},
        scale_factor=ia["diff_trainer_params"]["init_args"]["scale_factor"],
        inference={k: ia["inference_params"]["init_args"][k] for k in ("num_conditional_frames", "anchor_frames")},
        svd_pipeline_repo=mc["svd_pipeline"]["init_args"]["args"][0],
    )
    path = os.path.join(ROOT, "tests", "golden", "config_values.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, sort_keys=True)[:1500])
    print("wrote", path)


if __name__ == "__main__":
    main()
