"""What a 16-bit-operand MFMA execution can reach against the fp32 path AT BEST (build container or GPU box; CPU only):

    python oracle/measure_precision_floor.py [--arch tiny|full] [--dtype fp16|bf16]

Runs the fp32 oracle of StreamingWrapper.forward three ways on the same seeded case and prints per-frame L2 against the plain fp32 run:
  (1) every GEMM / convolution / attention OPERAND rounded to the 16-bit type (what a matrix core consumes), everything else -- the
      residual stream, every stored activation, norms, softmax, accumulation -- exact fp32.  This is the floor for ANY implementation that
      feeds 16-bit operands to MFMA, however it stores its activations (an fp32 residual stream included);
  (2) additionally every GEMM / conv / norm OUTPUT stored in 16 bit (residual adds in fp32 inside the epilogue, as the HIP kernels do).
Next to them DESIGN.md quotes oracle/measure_reference_autocast.py: the deviation of the REFERENCE'S OWN fp16 autocast from its fp32 path.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, svd_oracle as O  # noqa: E402
from streamingt2v_amd.params import init_by_name  # noqa: E402
from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="tiny")
    ap.add_argument("--dtype", default="fp16")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    DT = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    if a.arch == "tiny":
        tu = cases.TINY_UNET
        cfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"],
                         conditioning_embedding_out_channels=tu["cond_embed"])
        ocfg = O.Cfg(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"],
                     cond_embed_channels=tu["cond_embed"])
        inp, T, Tc, seeds = cases.tiny_wrapper_inputs(), tu["T"], tu["Tc"], (1, 2)
    else:
        cfg, ocfg, c = UNetConfig(), O.Cfg(), cases.FULLARCH_CASE
        inp, T, Tc, seeds = cases.fullarch_inputs(), c["T"], c["Tc"], (c["seed_unet"], c["seed_cn"])
    sd_u, sd_c = init_by_name(VideoUNet(cfg).spec(), seed=seeds[0]), init_by_name(ControlNet(cfg).spec(), seed=seeds[1])
    cond = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    run = lambda: O.streaming_wrapper(sd_u, sd_c, ocfg, inp["x"], inp["t"], cond, 2, T, Tc, inp["ctrl_frames"])
    ref = run()
    r = lambda t: t.to(DT).float() if t is not None and t.dtype == torch.float32 else t
    lin, c2, c3, sdpa, gn, ln = F.linear, F.conv2d, F.conv3d, F.scaled_dot_product_attention, F.group_norm, F.layer_norm

    def rep(name, out):
        e = (out - ref).flatten(1).pow(2).mean(1).sqrt()
        print(f"[{a.arch} architecture, {a.dtype}] {name}: per-frame L2 abs max {e.max():.3e} mean {e.mean():.3e}", flush=True)

    F.linear = lambda x, w, b=None: lin(r(x), r(w), b)
    F.conv2d = lambda x, w, b=None, *aa, **k: c2(r(x), r(w), b, *aa, **k)
    F.conv3d = lambda x, w, b=None, *aa, **k: c3(r(x), r(w), b, *aa, **k)
    F.scaled_dot_product_attention = lambda q, k, v, *aa, **kw: sdpa(r(q), r(k), r(v), *aa, **kw)
    rep("(1) 16-bit GEMM / conv / attention operands only, fp32 residual stream and activations", run())
    F.linear = lambda x, w, b=None: r(lin(r(x), r(w), b))
    F.conv2d = lambda x, w, b=None, *aa, **k: r(c2(r(x), r(w), b, *aa, **k))
    F.conv3d = lambda x, w, b=None, *aa, **k: r(c3(r(x), r(w), b, *aa, **k))
    F.group_norm = lambda x, *aa, **k: r(gn(x, *aa, **k))
    F.layer_norm = lambda x, *aa, **k: r(ln(x, *aa, **k))
    F.scaled_dot_product_attention = lambda q, k, v, *aa, **kw: r(sdpa(r(q), r(k), r(v), *aa, **kw))
    rep("(2) + 16-bit GEMM / conv / norm / attention outputs", run())


if __name__ == "__main__":
    main()
