#!/usr/bin/env python
"""In-situ GEMM tile autotuner (run on an MI355X):

    python tools/tune_gemm.py            # writes streamingt2v_amd/gemm_tiles.json

Runs one full-size forward of each workload (UNet w/ and w/o ControlNet+CAM, VAE decode) with a hook in ops.gemm:
the first time a GEMM signature (mode, M, N, K, stride, ups, epilogue, output kind) is seen, every valid tile config
is timed on the REAL operands of that call (HIP events on the launch stream, best of 3 after 1 warm-up) and the
fastest is recorded.  ops.gemm consults the resulting table; unknown signatures fall back to the C heuristic."""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from streamingt2v_amd import lib as L, ops  # noqa: E402

REPS = 3


class Tuner:
    def __init__(self):
        self.table = {}
        self.ncfg = L.lib.svd_gemm_num_configs()

    def select(self, args, stream):
        key = ops.gemm_signature(args)
        hit = self.table.get(key)
        if hit is not None:
            hit["calls"] += 1
            return hit["cfg"]
        res = {}
        flops = 2.0 * args.M * args.N * args.K
        for cfg in range(1, self.ncfg + 1):
            if cfg == 6 or L.lib.svd_gemm_config_valid(C.byref(args), cfg) != 1:
                continue
            args.tile_cfg = cfg
            if L.lib.svd_gemm(C.byref(args), stream) != 0:
                continue
            best = 1e9
            for _ in range(REPS):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                L.lib.svd_gemm(C.byref(args), stream)
                e.record()
                e.synchronize()
                best = min(best, s.elapsed_time(e))
            res[cfg] = best
        cfg = min(res, key=res.get)
        heur = L.lib.svd_gemm_pick_config(C.byref(args))
        self.table[key] = {"cfg": cfg, "calls": 1, "ms": round(res[cfg], 4), "tflops": round(flops / res[cfg] / 1e9, 1),
                           "heuristic_cfg": heur, "heuristic_ms": round(res.get(heur, float("nan")), 4),
                           "all_ms": {str(k): round(v, 4) for k, v in sorted(res.items())}}
        args.tile_cfg = 0
        return cfg


def main():
    import argparse
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="'ar16' / 'ar': the AR forward + decode only, under the all-16-bit plan / the default precision plan (A/B runs of kernel variants)")
    ap.add_argument("--out", default="", help="write the table here instead of streamingt2v_amd/gemm_tiles.json")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    tuner = Tuner()
    ops.tuner = tuner
    ops._tile_table = {}
    t0 = time.time()
    # both residual-stream forms of the UNet / ControlNet: their producers are different kernels (fp32 stream: `_o1` signatures)
    plans = {"ar16": (("ar_chunk", False),), "ar": (("ar_chunk", True),)}.get(a.only, (("ar_chunk", True), ("c2", True), ("ar_chunk", False), ("c2", False)))
    for workload, stream in plans:
        ops.set_stream_f32(stream)                      # True: every block's stream in fp32 (= the round-4 default plan); False: the all-16-bit plan
        ops.set_precision_plan(exact_rim=True, cn_stream_f32=stream, stream_f32_min_ch=320 if stream else 0)
        wrapper, vae = bench.build_models(workload, dev)
        from streamingt2v_amd.sampling import EulerEDMSampler
        from streamingt2v_amd.streaming_svd import StreamingSVD
        model = StreamingSVD(wrapper, vae, EulerEDMSampler(num_steps=1, num_frames=25))
        c, uc, ctrl, noise = bench.synthetic_inputs(dev, 33)
        with torch.no_grad():
            model._generate_conditional_output(c, uc, ctrl if workload == "ar_chunk" else None, noise)
        torch.cuda.synchronize()
        print(f"{workload} (residual stream {'fp32' if stream else '16 bit'}): {len(tuner.table)} signatures after {time.time() - t0:.0f}s", flush=True)
        del wrapper, vae, model
        torch.cuda.empty_cache()
    ops.set_stream_f32(False)
    ops.set_precision_plan(exact_rim=True, cn_stream_f32=True, stream_f32_min_ch=320)
    if not a.only:
        # enhancement stage: one I2VGen-XL UNet forward of a 38-frame window (CFG batch 2) at latent 90x160
        from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
        from streamingt2v_amd.params import init_by_name
        unet = I2VGenXLUNet(I2VConfig())
        unet.load_state_dict(init_by_name(unet.spec(), seed=5, device=dev), device=dev)
        g = torch.Generator(device=dev); g.manual_seed(1)
        rn = lambda *sh: torch.randn(*sh, generator=g, device=dev)
        with torch.no_grad():
            unet(rn(2, 4, 38, 90, 160), 500, fps=torch.tensor([16, 16]), image_latents=rn(2, 4, 38, 90, 160), image_embeddings=rn(2, 1024),
                 encoder_hidden_states=rn(2, 77, 1024))
        torch.cuda.synchronize()
        print(f"enhance: {len(tuner.table)} signatures after {time.time() - t0:.0f}s", flush=True)
        del unet
        torch.cuda.empty_cache()
    out = a.out or os.path.join(ROOT, "streamingt2v_amd", "gemm_tiles.json")
    gain = sum(v["heuristic_ms"] - v["ms"] for v in tuner.table.values() if v["heuristic_ms"] == v["heuristic_ms"])
    with open(out, "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "note": "best tile config per GEMM signature, tools/tune_gemm.py",
                   "table": dict(sorted(tuner.table.items()))}, f, indent=0)
    print(f"wrote {out}: {len(tuner.table)} signatures; summed per-signature gain over heuristic {gain:.1f} ms")
    rank = sorted(tuner.table.items(), key=lambda kv: -kv[1]["calls"] * kv[1]["ms"])
    tot = sum(v["calls"] * v["ms"] for v in tuner.table.values())
    print(f"GEMM time of the two tuned forwards + decode: {tot:.1f} ms; top signatures (calls x ms):")
    for k, v in rank[:45]:
        print(f"  {v['calls'] * v['ms']:8.2f} ms  {v['calls']:4d} x {v['ms']:7.3f}  cfg{v['cfg']:<3d} {v['tflops']:6.0f} TF  {k}")
    if not a.out and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        import shutil
        shutil.copy(out, os.path.join(ROOT, "gpurun_out", "gemm_tiles.json"))


if __name__ == "__main__":
    main()
