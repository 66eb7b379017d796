"""Pin the enhancer's CALL-level logic against the reference's own code (build container only).

    python oracle/make_golden_i2v_pipeline.py      # needs /root/reference ; writes tests/golden/i2v_call_tiny.pt

Runs the UNMODIFIED `I2VGenXLPipeline.__call__` (code/i2v_enhance/pipeline_i2vgen_xl.py:607-935) on CPU through
oracle/i2v_pipeline_bootstrap.py -- vendored tiny I2VGenXLUNet, linear stand-ins for VAE / CLIP image encoder, restated diffusers
plumbing -- on the seeded case of oracle/cases.tiny_i2v_call_inputs (2 key images, 10 frames, 2 blending windows with overlap 2, CFG 9,
3 DDIM steps), records what it feeds the UNet (per-window conditioning, initial noisy latents, timesteps) and its final latents, and
asserts that the restatement oracle/i2vgen_oracle.enhance_call reproduces all of it.  The recorded tensors are the golden vectors the
CPU suite checks the oracle and the product's host side (enhance_codec.EnhanceCodec) against.
"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import i2v_pipeline_bootstrap as pb  # noqa: E402

pb.install()
from oracle import i2vgen_oracle as O  # noqa: E402
from oracle.cases import TINY_I2V, TINY_I2V_CALL, tiny_i2v_call_inputs, tiny_i2v_kwargs  # noqa: E402
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    from i2v_enhance.unet_i2vgen_xl import I2VGenXLUNet
    c = TINY_I2V_CALL
    unet = I2VGenXLUNet(**tiny_i2v_kwargs()).eval()
    spec = Spec()
    for k, v in unet.state_dict().items():
        spec.add(k, *v.shape)
    sd = init_by_name(spec, seed=5)
    unet.load_state_dict(sd, strict=True)
    pipe, mod = pb.build_pipeline(unet, TINY_I2V["cross_attention_dim"])
    inp = tiny_i2v_call_inputs()

    calls = []
    real_forward = unet.forward

    def recording_forward(sample, t, **kw):
        calls.append(dict(sample=sample.clone(), t=int(t), fps=kw["fps"].clone(), image_latents=kw["image_latents"].clone(),
                          image_embeddings=kw["image_embeddings"].clone(), text=kw["encoder_hidden_states"].clone()))
        return real_forward(sample, t, **kw)

    unet.forward = recording_forward
    kw = dict(height=c["H"], width=c["W"], strength=c["strength"], overlap_size=c["overlap"], chunk_size=c["chunk"], num_frames=c["chunk"],
              num_inference_steps=c["steps"], guidance_scale=c["guidance"])
    torch.manual_seed(777)                        # prepare_image_latents samples WITHOUT the generator (:486): global stream
    random.seed(c["py_seed"])                     # the blending offsets come from the global `random` module (:894)
    ref = pipe(prompt=None, image=inp["images"], video=inp["frames"], prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
               generator=torch.Generator().manual_seed(c["gen_seed"]), output_type="latent", return_dict=False, **kw)[0]
    assert len(calls) == 2 * 3 and [x["t"] for x in calls[::2]] == [x["t"] for x in calls[1::2]]

    trace = {}
    torch.manual_seed(777)
    ora = O.enhance_call(sd, inp["images"], inp["frames"], inp["prompt_embeds"], inp["negative_prompt_embeds"], pb.FakeVAE(),
                         pb.FakeImageEncoder(TINY_I2V["cross_attention_dim"]), torch.Generator().manual_seed(c["gen_seed"]), random.Random(c["py_seed"]),
                         height=c["H"], width=c["W"], chunk_size=c["chunk"], overlap_size=c["overlap"], num_inference_steps=c["steps"],
                         strength=c["strength"], guidance_scale=c["guidance"], trace=trace)
    errs = dict(final=(ref - ora).abs().max().item(), init=(calls[0]["sample"][:1] - trace["init_latents"][:, :, : c["chunk"]]).abs().max().item())
    for idx in range(2):
        errs[f"image_latents{idx}"] = (calls[idx]["image_latents"] - trace["image_latents"][idx]).abs().max().item()
        errs[f"image_embeddings{idx}"] = (calls[idx]["image_embeddings"] - trace["image_embeddings"][idx]).abs().max().item()
        assert torch.equal(calls[idx]["fps"], trace["fps"]) and torch.equal(calls[idx]["text"], trace["text"])
    assert [x["t"] for x in calls[::2]] == trace["timesteps"], ([x["t"] for x in calls[::2]], trace["timesteps"])
    print("[i2v call] reference-vs-oracle max abs err:", {k: f"{v:.2e}" for k, v in errs.items()}, "| timesteps", trace["timesteps"],
          "| fps", trace["fps"].tolist(), f"| final std {ref.std():.3f}")
    assert max(errs.values()) <= 1e-4, errs
    # encode_prompt (:169-347) with the fork's default clip_skip = 1, on a real (tiny, random) transformers CLIPTextModel: the oracle's
    # text tower with clip_skip must reproduce it; the defaults of __call__ that i2v_enhance_interface.py relies on are recorded.
    import inspect
    import types
    import transformers as tr
    from oracle.clip_text_oracle import text_tower
    torch.manual_seed(0)
    tcfg = tr.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                             max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5, eos_token_id=2, bos_token_id=0, pad_token_id=1)
    hf = tr.CLIPTextModel(tcfg).eval()
    for p_ in hf.parameters():
        p_.normal_(0, 0.05)
    if not hasattr(hf, "text_model"):
        object.__setattr__(hf, "text_model", hf)                     # transformers 4.40 layout (the reference's pin)
    ids = {"P": torch.randint(3, 1000, (1, 77)), "N": torch.randint(3, 1000, (1, 77))}

    class Tok:
        model_max_length = 77

        def __call__(self, text, **kw):
            t = text[0] if isinstance(text, list) else text
            return types.SimpleNamespace(input_ids=ids[t], attention_mask=torch.ones_like(ids[t]))

        def batch_decode(self, x):
            return [""]
    defaults = {k: v.default for k, v in inspect.signature(mod.I2VGenXLPipeline.__call__).parameters.items()
                if isinstance(v.default, (int, float, str, bool, type(None))) and v.default is not inspect.Parameter.empty}
    bare = types.SimpleNamespace(tokenizer=Tok(), text_encoder=hf, unet=None, do_classifier_free_guidance=True)
    pe, ne = mod.I2VGenXLPipeline.encode_prompt(bare, "P", "cpu", 1, "N", clip_skip=defaults["clip_skip"])
    tsd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in hf.state_dict().items() if "position_ids" not in k}
    e = max((pe - text_tower(tsd, ids["P"], 2, clip_skip=defaults["clip_skip"])).abs().max().item(),
            (ne - text_tower(tsd, ids["N"], 2, clip_skip=defaults["clip_skip"])).abs().max().item())
    print(f"[i2v call] encode_prompt(clip_skip={defaults['clip_skip']}) reference-vs-oracle {e:.2e}; __call__ defaults: target_fps {defaults['target_fps']}, "
          f"clip_skip {defaults['clip_skip']}, decode_chunk_size {defaults['decode_chunk_size']}")
    assert e <= 1e-5
    out = os.path.join(ROOT, "tests", "golden", "i2v_call_tiny.pt")
    torch.save(dict(final=ref.clone(), init_latents=trace["init_latents"], clean=trace["clean"], noise=trace["noise"], timesteps=trace["timesteps"], fps=calls[0]["fps"],
                    image_latents=[calls[i]["image_latents"] for i in range(2)], image_embeddings=[calls[i]["image_embeddings"] for i in range(2)],
                    text=calls[0]["text"], vae_encode_calls=pipe.vae.encode_calls, call_defaults=defaults), out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
