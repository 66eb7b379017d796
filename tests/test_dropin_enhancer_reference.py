"""CPU suite: the ENHANCER-side swap of INTEGRATION.md (round-4 review: "the one boundary that has not been executed"), EXECUTED.

The reference's UNMODIFIED `I2VGenXLPipeline.__call__` (code/i2v_enhance/pipeline_i2vgen_xl.py:617-930: SDEdit start, CFG batching, the denoise
loop with randomized blending :841-913, `self.unet(...)` at :857-867) runs on CPU through oracle/i2v_pipeline_bootstrap.py around OUR
`I2VGenXLUNet`, installed by `streamingt2v_amd.dropin.install_enhancer` -- the function INTEGRATION.md tells a maintainer to call -- with the HIP
launchers replaced by the fp32 torch statements of tests/svd_shim.py.  The final latents must equal the golden the same pipeline produced around
the reference's own vendored UNet (tests/golden/i2v_call_tiny.pt, oracle/make_golden_i2v_pipeline.py).
Needs /root/reference (build container); skipped elsewhere."""
import os
import random

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "i2v_call_tiny.pt")
needs_ref = pytest.mark.skipif(not os.path.isdir("/root/reference/code"), reason="the unmodified reference is only present in the build container")


@needs_ref
@pytest.mark.parametrize("plan", ["default", "16bit"])
def test_reference_pipeline_call_drives_our_enhancer_unet(monkeypatch, plan):
    from oracle import i2v_pipeline_bootstrap as pb
    pb.install()
    from i2v_enhance.unet_i2vgen_xl import I2VGenXLUNet as RefUNet
    from oracle.cases import TINY_I2V, TINY_I2V_CALL, tiny_i2v_call_inputs, tiny_i2v_kwargs
    from streamingt2v_amd import dropin, ops
    from streamingt2v_amd.params import Spec, init_by_name
    from tests import svd_shim
    torch.set_grad_enabled(False)
    svd_shim.install(monkeypatch)
    if plan == "16bit":
        monkeypatch.setattr(ops, "I2V_EXACT_RIM", False)
        monkeypatch.setattr(ops, "I2V_STREAM_F32_MIN_CH", 0)
    c, gold = TINY_I2V_CALL, torch.load(GOLD)
    ref_unet = RefUNet(**tiny_i2v_kwargs()).eval()                      # what I2VGenXLPipeline.from_pretrained would have put there
    spec = Spec()
    for k, v in ref_unet.state_dict().items():
        spec.add(k, *v.shape)
    ref_unet.load_state_dict(init_by_name(spec, seed=5), strict=True)
    pipe, _ = pb.build_pipeline(ref_unet, TINY_I2V["cross_attention_dim"])
    ours = dropin.install_enhancer(pipe, device="cpu")                   # <- the swap of INTEGRATION.md
    assert isinstance(pipe.unet, dropin.HipModule) and pipe.unet.impl is ours and pipe.unet.config.in_channels == 4
    calls = []
    fwd = ours.forward
    ours.forward = lambda *a, **k: (calls.append(sorted(k)), fwd(*a, **k))[1]
    inp = tiny_i2v_call_inputs()
    kw = dict(height=c["H"], width=c["W"], strength=c["strength"], overlap_size=c["overlap"], chunk_size=c["chunk"], num_frames=c["chunk"],
              num_inference_steps=c["steps"], guidance_scale=c["guidance"])
    torch.manual_seed(777)
    random.seed(c["py_seed"])
    out = pipe(prompt=None, image=inp["images"], video=inp["frames"], prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
               generator=torch.Generator().manual_seed(c["gen_seed"]), output_type="latent", return_dict=False, **kw)[0]
    assert len(calls) == 2 * len(gold["timesteps"]), len(calls)           # 2 blending windows x 3 SDEdit DDIM steps, each one batched CFG call
    assert {"encoder_hidden_states", "fps", "image_latents", "image_embeddings", "return_dict"} <= set(calls[0])
    e = (out - gold["final"]).abs().max().item()
    print(f"[executed enhancer drop-in, plan {plan}] final latents vs the all-reference run: max abs {e:.3e} (|final| std {gold['final'].std():.3f})")
    assert e <= 2e-4, e            # fp32 on both sides: summation order through 3 CFG-9 DDIM steps (measured 2.3e-5; the oracle restatement agrees to 1.6e-5)
