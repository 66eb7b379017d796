"""tests/golden/i2v_fullarch.pt: the reference's UNMODIFIED vendored I2VGenXLUNet at its SHIPPED configuration (block_out_channels
320-640-1280-1280, 2 layers per block, cross-attention dim 1024; 1.42 B parameters) on a small latent (build container only).

    python oracle/make_golden_i2v_fullarch.py      # ~15 GB of RAM, a few minutes
    python oracle/make_golden_i2v_fullarch.py --fullres   # round 5: the same network at the SHIPPED latent size 90 x 160 (720 x 1280 pixels),
                                                          # CFG 2 x 4 frames, N = 14 400 spatial attention -> tests/golden/i2v_fullres.pt (1.8 MB)

The tiny golden (make_golden_i2v.py) has 3 levels and 1 layer per block; this pins the oracle's wiring of the real 4-level network (skip
connections with forwarded upsample sizes on the odd 9 x 16 latent, temporal layers at every level, 145-token context) and stores the
reference output for tools/fullarch_parity_i2v.py.
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
from oracle.cases import I2V_FULLARCH_CASE, I2V_FULLRES_CASE, i2v_fullarch_inputs, i2v_fullres_inputs  # noqa: E402
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    from i2v_enhance.unet_i2vgen_xl import I2VGenXLUNet
    t0 = time.time()
    unet = I2VGenXLUNet().eval()
    spec = Spec()
    for k, v in unet.state_dict().items():
        spec.add(k, *v.shape)
    sd = init_by_name(spec, seed=I2V_FULLARCH_CASE["seed"])
    unet.load_state_dict(sd, strict=True)
    print(f"vendored I2VGenXLUNet built and loaded in {time.time() - t0:.0f} s ({sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters)")
    fullres = "--fullres" in sys.argv
    inp = i2v_fullres_inputs() if fullres else i2v_fullarch_inputs()
    t0 = time.time()
    ref = unet(inp["sample"], inp["t"], fps=inp["fps"], image_latents=inp["image_latents"], image_embeddings=inp["image_embeddings"],
               encoder_hidden_states=inp["text"], return_dict=False)[0]
    t1 = time.time()
    ora = O.unet(sd, inp["sample"], inp["t"], inp["fps"], inp["image_latents"], inp["image_embeddings"], inp["text"])
    e = (ref - ora).abs().max().item()
    print(f"[i2v full architecture] vendored-vs-oracle max abs err {e:.3e} (|ref| std {ref.std():.3f}); reference {t1 - t0:.0f} s, oracle {time.time() - t1:.0f} s")
    assert e <= 5e-4, e
    path = os.path.join(ROOT, "tests", "golden", "i2v_fullres.pt" if fullres else "i2v_fullarch.pt")
    torch.save({"out": ref.clone(), "cpu_seconds": t1 - t0, "oracle_max_abs_err": e}, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
