"""Generate tests/golden/i2v_tiny.pt from the UNMODIFIED vendored I2VGen-XL UNet of the reference (build container only).

    python oracle/make_golden_i2v.py        # needs /root/reference ; writes tests/golden/i2v_tiny.pt

The vendored modules (code/i2v_enhance/*.py) are imported through oracle/i2v_bootstrap.py, whose fake ``diffusers`` package
restates the diffusers==0.30.2 leaf layers (diffusers itself is not available: parity unpinned for the leaves).  Weights are
by-name deterministic (streamingt2v_amd.params.init_by_name; load_state_dict(strict=True) proves the key/shape spec), inputs
are seeded (oracle/cases.py).  The restatement oracle/i2vgen_oracle.py must agree with the vendored forward to <= 2e-4.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import i2v_bootstrap  # noqa: E402

i2v_bootstrap.install()
from oracle import i2vgen_oracle as O  # noqa: E402
from oracle.cases import tiny_i2v_inputs, tiny_i2v_kwargs  # noqa: E402
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TOL = 2e-4


def spec_of(module):
    s = Spec()
    for k, v in module.state_dict().items():
        s.add(k, *v.shape)
    return s


def main():
    torch.set_grad_enabled(False)
    from i2v_enhance.unet_i2vgen_xl import I2VGenXLUNet
    t0 = time.time()
    unet = I2VGenXLUNet(**tiny_i2v_kwargs()).eval()
    sd = init_by_name(spec_of(unet), seed=5)
    unet.load_state_dict(sd, strict=True)
    inp = tiny_i2v_inputs()
    ref = unet(inp["sample"], inp["t"], fps=inp["fps"], image_latents=inp["image_latents"], image_embeddings=inp["image_embeddings"],
               encoder_hidden_states=inp["text"], return_dict=False)[0]
    ref_memopt = unet(inp["sample"], inp["t"], fps=inp["fps"], image_latents=inp["image_latents"],
                      image_embeddings=inp["image_embeddings"], encoder_hidden_states=inp["text"], return_dict=False, use_memopt=True)[0]
    print(f"[i2v] use_memopt changes the output by {(ref - ref_memopt).abs().max().item():.2e} (chunking is per batch element)")
    ora = O.unet(sd, inp["sample"], inp["t"], inp["fps"], inp["image_latents"], inp["image_embeddings"], inp["text"])
    e = (ref - ora).abs().max().item()
    print(f"[i2v unet] vendored-vs-oracle max abs err {e:.3e} (|ref| max {ref.abs().max():.3f}, std {ref.std():.3f})  {time.time() - t0:.1f}s")
    assert e <= TOL, e
    # our host-side spec must equal the vendored module's state_dict (tiny and full size)
    try:
        from streamingt2v_amd.i2vgen_unet import I2VGenXLUNet as Ours, I2VConfig
        kw = tiny_i2v_kwargs()
        assert dict(Ours(I2VConfig(block_out_channels=kw["block_out_channels"], layers_per_block=1, cross_attention_dim=kw["cross_attention_dim"],
                                   attn_levels=(True, True, False))).spec()) == dict(spec_of(unet)), "tiny spec mismatch"
        with torch.device("meta"):
            full = I2VGenXLUNet()
        assert dict(Ours(I2VConfig()).spec()) == dict(spec_of(full)), "full-size spec mismatch"
        print("host-side I2VGenXLUNet specs equal the vendored state_dict (tiny: %d tensors, full: %d tensors, %.1f M params)"
              % (len(spec_of(unet)), len(spec_of(full)), sum(v.numel() for v in full.state_dict().values()) / 1e6))
    except ImportError as ex:
        print("host-side spec check skipped:", ex)
    os.makedirs(OUT, exist_ok=True)
    torch.save({"out": ref.clone()}, os.path.join(OUT, "i2v_tiny.pt"))
    print("wrote", os.path.join(OUT, "i2v_tiny.pt"))


if __name__ == "__main__":
    main()
