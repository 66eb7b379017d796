"""The enhancer's CALL-level logic (everything around the UNet in I2VGenXLPipeline.__call__, pipeline_i2vgen_xl.py:607-935) on CPU, against
tests/golden/i2v_call_tiny.pt = what the reference's UNMODIFIED __call__ fed its (vendored, tiny) UNet and what it returned
(oracle/make_golden_i2v_pipeline.py; VAE / CLIP image encoder replaced on both sides by the linear stand-ins of
oracle/i2v_pipeline_bootstrap.py):
  * the oracle restatement `i2vgen_oracle.enhance_call` reproduces the reference's final latents;
  * the PRODUCT's host side -- enhance_codec.EnhanceCodec: crops, CLIP preprocessing, frame-position planes, CFG batching, fps, chunked video
    encoding, order and shape of the random draws -- reproduces the reference's UNet inputs from the same seeds."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import i2vgen_oracle as O  # noqa: E402
from oracle.cases import TINY_I2V, TINY_I2V_CALL, tiny_i2v_call_inputs  # noqa: E402
from oracle.i2v_pipeline_bootstrap import FakeImageEncoder, FakeVAE  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "i2v_call_tiny.pt")


def _tiny_sd():
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    from oracle.cases import tiny_i2v_kwargs
    kw = tiny_i2v_kwargs()
    u = I2VGenXLUNet(I2VConfig(block_out_channels=kw["block_out_channels"], layers_per_block=kw["layers_per_block"],
                               cross_attention_dim=kw["cross_attention_dim"], attn_levels=(True, True, False)))
    return init_by_name(u.spec(), seed=5)


def test_oracle_enhance_call_matches_reference_call():
    torch.set_grad_enabled(False)
    c, inp, g = TINY_I2V_CALL, tiny_i2v_call_inputs(), torch.load(GOLD)
    trace = {}
    torch.manual_seed(777)
    out = O.enhance_call(_tiny_sd(), inp["images"], inp["frames"], inp["prompt_embeds"], inp["negative_prompt_embeds"], FakeVAE(),
                         FakeImageEncoder(TINY_I2V["cross_attention_dim"]), torch.Generator().manual_seed(c["gen_seed"]), random.Random(c["py_seed"]),
                         height=c["H"], width=c["W"], chunk_size=c["chunk"], overlap_size=c["overlap"], num_inference_steps=c["steps"],
                         strength=c["strength"], guidance_scale=c["guidance"], trace=trace)
    assert trace["timesteps"] == g["timesteps"] and torch.equal(trace["init_latents"], g["init_latents"])
    assert (out - g["final"]).abs().max() <= 1e-4


class _VaeAdapter:
    """The stand-in VAE behind the product's AutoencoderKL2D interface (encode_sample = posterior sample x scaling factor)."""

    def __init__(self):
        self.f = FakeVAE()

    def encode_sample(self, x, generator=None):
        m = self.f.mean(x)
        return (m + self.f.std * torch.randn(m.shape, generator=generator)) * self.f.config.scaling_factor


def test_product_host_side_reproduces_reference_unet_inputs():
    from streamingt2v_amd.enhance import DDIMSchedule
    from streamingt2v_amd.enhance_codec import EnhanceCodec
    torch.set_grad_enabled(False)
    c, inp, g = TINY_I2V_CALL, tiny_i2v_call_inputs(), torch.load(GOLD)
    tower = FakeImageEncoder(TINY_I2V["cross_attention_dim"])
    codec = EnhanceCodec(_VaeAdapter(), tower.embed, None, height=c["H"], width=c["W"], generator=torch.Generator().manual_seed(c["gen_seed"]), device="cpu")
    assert codec.fps == c["fps"] == 38                                       # the fork's target_fps default (:630), not diffusers' 16
    codec.set_prompt_embeds(inp["prompt_embeds"], inp["negative_prompt_embeds"])
    torch.manual_seed(777)
    conds = codec.window_conditioning(inp["images"], 2, c["chunk"])           # same order of random draws as the reference: image latents,
    lat = codec.encode_video(inp["frames"])                                   # video posterior sample,
    noise = codec.noise_like(lat)                                             # SDEdit noise
    for i in range(2):
        assert torch.equal(conds[i]["fps"], g["fps"]) and torch.equal(conds[i]["text"], g["text"])
        assert (conds[i]["image_latents"] - g["image_latents"][i]).abs().max() <= 2e-6
        assert (conds[i]["image_embeddings"] - g["image_embeddings"][i].squeeze(1)).abs().max() <= 2e-5
    assert (lat - g["clean"]).abs().max() <= 2e-6 and torch.equal(noise, g["noise"])
    sched = DDIMSchedule()
    assert sched.get_timesteps(c["steps"], c["strength"]) == g["timesteps"]
    assert (sched.add_noise(lat, noise, g["timesteps"][0]) - g["init_latents"]).abs().max() <= 2e-6


def test_product_defaults_match_reference_call_defaults():
    """Arguments i2v_enhance_interface.py does NOT pass take the defaults of the fork's __call__ (recorded in the golden): the product must
    use the same ones -- fps conditioning 38, clip_skip 1 (last text-encoder layer skipped), frame-by-frame decoding."""
    import inspect
    from streamingt2v_amd.enhance_codec import EnhanceCodec
    from streamingt2v_amd.pipeline import DEFAULTS
    d = torch.load(GOLD)["call_defaults"]
    assert (d["target_fps"], d["clip_skip"], d["decode_chunk_size"], d["eta"]) == (38, 1, 1, 0.0)
    assert inspect.signature(EnhanceCodec.__init__).parameters["target_fps"].default == d["target_fps"] == DEFAULTS["enhance_target_fps"]
    assert inspect.signature(EnhanceCodec.set_prompts_from_ids).parameters["clip_skip"].default == d["clip_skip"]


def test_encode_video_chunks_like_torch_chunk():
    """prepare_video_latents :585-597: more than 16 frames are encoded as torch.chunk(video, F // 16) -- ceil-sized chunks, last one short."""
    from streamingt2v_amd.enhance_codec import EnhanceCodec
    import numpy as np
    for n in (10, 16, 17, 38, 100, 47):
        vae = _VaeAdapter()
        sizes = []
        orig = vae.encode_sample
        vae.encode_sample = lambda x, generator=None: (sizes.append(x.shape[0]), orig(x, generator))[1]
        codec = EnhanceCodec(vae, None, None, height=16, width=32, generator=torch.Generator().manual_seed(0), device="cpu")
        out = codec.encode_video([np.zeros((16, 32, 3), dtype=np.uint8)] * n)
        want = [t.shape[0] for t in torch.chunk(torch.zeros(n, 1), n // 16, 0)] if n > 16 else [n]
        assert sizes == want and out.shape == (1, 4, n, 2, 4), (n, sizes, want)
