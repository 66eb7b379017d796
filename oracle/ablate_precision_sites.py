"""Which SITES of StreamingWrapper.forward carry the 16-bit deviation from the fp32 path?  (TEST INFRASTRUCTURE, CPU only.)

    python oracle/ablate_precision_sites.py [--arch tiny|full] [--mode zones|kinds|cumulative]

The fp32 oracle runs with the emulation (2) of oracle/measure_precision_floor.py -- every GEMM / conv / attention operand AND every GEMM /
conv / norm / attention output rounded to fp16, residual adds in fp32 -- except inside one ZONE of the network at a time, which stays exact
fp32.  The drop of the per-frame L2 against the plain fp32 run is that zone's share of the error budget (the shares add in quadrature).
Zones follow the reference's module tree (models/diffusion/video_model.py:540-618): controlnet | stem | input_blocks.N | middle | mergers |
output_blocks.N | head (out.0 + out.2).  `kinds` keeps one KIND of rounding exact everywhere instead (operands / outputs of linear, conv,
norm, attention), `cumulative` keeps the LAST k output blocks + head exact (the candidates for a higher-precision tail on the GPU).
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

ZONE = ["?"]          # current zone (a stack of one)
NET = ["unet"]


def zone_of(prefix):
    parts = prefix.split(".")
    if NET[0] == "controlnet":
        if parts[0] == "input_blocks":
            return "controlnet.input_blocks." + parts[1]
        return "controlnet." + ("middle" if parts[0].startswith("middle_block") else parts[0])
    if parts[0] in ("input_blocks", "output_blocks"):
        return parts[0] + "." + parts[1]
    if parts[0].startswith("middle_block"):
        return "middle"
    if parts[0].startswith("cross_attention_merger"):
        return "mergers"
    return parts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="full")
    ap.add_argument("--mode", default="zones")
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

    # ---- zone tracking: wrap the oracle's block functions (their second argument is the state-dict prefix) ----
    def zoned(fn):
        def w(sd, p, *aa, **kw):
            old = ZONE[0]
            ZONE[0] = zone_of(p)
            try:
                return fn(sd, p, *aa, **kw)
            finally:
                ZONE[0] = old
        return w
    for name in ("video_res_block", "spatial_video_transformer", "conditional_model"):
        setattr(O, name, zoned(getattr(O, name)))
    cn0 = O.controlnet

    def cn(*aa, **kw):
        NET[0] = "controlnet"; ZONE[0] = "?"
        try:
            return cn0(*aa, **kw)
        finally:
            NET[0] = "unet"; ZONE[0] = "?"
    O.controlnet = cn
    # the stem / down / up convs and the head are called from _encoder / video_unet directly: zone by weight identity
    wzone = {}
    for k, v in sd_c.items():
        if k.startswith("controlnet_cond_embedding."):
            wzone[id(v)] = "controlnet.cond_embed"
        elif k.startswith("input_blocks.0.0."):
            wzone[id(v)] = "controlnet.stem"
        elif ".op." in k:
            wzone[id(v)] = "controlnet." + ".".join(k.split(".")[:2])
        elif k.startswith("time_embed") or k.startswith("label_emb"):
            wzone[id(v)] = "controlnet.emb"
    for k, v in sd_u.items():
        if k.startswith("input_blocks.0.0."):
            wzone[id(v)] = "stem"
        elif k.startswith("out."):
            wzone[id(v)] = "head"
        elif ".op." in k or ".conv." in k:
            wzone[id(v)] = zone_of(k)

    exact_zones, exact_kinds = set(), set()
    rules = []            # combos: (zone prefix, set of kinds or None = every kind) -> exact
    seen = {}

    def zone_now(w=None):
        if ZONE[0] == "?" and w is not None:
            return wzone.get(id(w), "emb")
        return ZONE[0]

    def r(t, kind, z):
        seen[z] = seen.get(z, 0) + 1
        if z in exact_zones or kind in exact_kinds or any(z.startswith(zp) and (ks is None or kind in ks) for zp, ks in rules):
            return t
        return t.to(DT).float() if t is not None and t.dtype == torch.float32 else t

    lin, c2, c3, sdpa, gn, ln = F.linear, F.conv2d, F.conv3d, F.scaled_dot_product_attention, F.group_norm, F.layer_norm

    def f_lin(x, w, b=None):
        z = zone_now(w)
        return r(lin(r(x, "lin_in", z), r(w, "lin_w", z), b), "lin_out", z)

    def f_c2(x, w, b=None, *aa, **k):
        z = zone_now(w)
        return r(c2(r(x, "conv_in", z), r(w, "conv_w", z), b, *aa, **k), "conv_out", z)

    def f_c3(x, w, b=None, *aa, **k):
        z = zone_now(w)
        return r(c3(r(x, "conv_in", z), r(w, "conv_w", z), b, *aa, **k), "conv_out", z)

    def f_gn(x, g, w=None, b=None, eps=1e-5):
        z = zone_now(w)
        return r(gn(x, g, w, b, eps), "norm_out", z)

    def f_ln(x, shape, w=None, b=None, eps=1e-5):
        z = zone_now(w)
        return r(ln(x, shape, w, b, eps), "norm_out", z)

    round_p = [False]

    def f_sdpa(q, k, v, *aa, **kw):
        z = zone_now()
        q, k, v = r(q, "attn_in", z), r(k, "attn_in", z), r(v, "attn_in", z)
        if round_p[0]:
            # what the MFMA attention kernels do on top: the probabilities are a 16-bit MFMA operand of P V (un-normalised: exp(s - max) in (0, 1],
            # the row sum divides the fp32 accumulator afterwards)
            sc = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
            pmat = torch.exp(sc - sc.amax(-1, keepdim=True))
            o = (r(pmat, "attn_p", z) @ v) / pmat.sum(-1, keepdim=True)
            return r(o, "attn_out", z)
        return r(sdpa(q, k, v, *aa, **kw), "attn_out", z)

    F.linear, F.conv2d, F.conv3d, F.group_norm, F.layer_norm, F.scaled_dot_product_attention = f_lin, f_c2, f_c3, f_gn, f_ln, f_sdpa

    def err():
        e = (run() - ref).flatten(1).pow(2).mean(1).sqrt()
        return float(e.mean()), float(e.max())

    base = err()
    print(f"[{a.arch}, {a.dtype}] emulation (2), nothing exact: mean {base[0]:.3e} max {base[1]:.3e}", flush=True)
    def zkey(z):
        return [(0, int(t)) if t.isdigit() else (1, t) for t in z.split(".")]
    zones = sorted(seen, key=zkey)
    print("zones:", {z: seen[z] for z in zones}, flush=True)
    if a.mode == "zones":
        for z in zones:
            exact_zones.clear(); exact_zones.add(z)
            m, x = err()
            print(f"  exact {z:18s}: mean {m:.3e} max {x:.3e}   share of mean^2 {1 - (m / base[0]) ** 2:6.1%}", flush=True)
    elif a.mode == "kinds":
        groups = {"all outputs (= operand-only floor)": {"lin_out", "conv_out", "norm_out", "attn_out"}, "norm outputs": {"norm_out"},
                  "linear outputs": {"lin_out"}, "conv outputs": {"conv_out"}, "attention outputs": {"attn_out"},
                  "attention operands": {"attn_in"}, "weights": {"lin_w", "conv_w"}, "activation operands": {"lin_in", "conv_in"},
                  "conv operands+weights": {"conv_in", "conv_w"}, "linear operands+weights": {"lin_in", "lin_w"}}
        for name, ks in groups.items():
            exact_kinds.clear(); exact_kinds.update(ks)
            m, x = err()
            print(f"  exact {name:36s}: mean {m:.3e} max {x:.3e}   share of mean^2 {1 - (m / base[0]) ** 2:6.1%}", flush=True)
    elif a.mode == "attnp":
        round_p[0] = True
        m, x = err()
        print(f"  + attention probabilities rounded to {a.dtype} before P V (the kernels' extra operand rounding): mean {m:.3e} max {x:.3e}   "
              f"mean^2 grows by {(m / base[0]) ** 2 - 1:6.1%}", flush=True)
        OUT = {"lin_out", "conv_out", "norm_out", "attn_out"}
        for name, rl in {"rim (cond_embed, stems, head, emb) exact": [("controlnet.cond_embed", None), ("controlnet.stem", None), ("controlnet.emb", None), ("head", None), ("stem", None), ("emb", None)],
                         "rim exact + ControlNet outputs exact": [("controlnet.cond_embed", None), ("controlnet.stem", None), ("controlnet.emb", None), ("head", None), ("stem", None), ("emb", None), ("controlnet", OUT)],
                         "rim + CN outputs + UNet outputs at >= 640 channels exact": [("controlnet.cond_embed", None), ("controlnet.stem", None), ("controlnet.emb", None), ("head", None), ("stem", None), ("emb", None), ("controlnet", OUT)]
                         + [(z, OUT) for z in ["input_blocks.%d" % i for i in range(4, 12)] + ["middle"] + ["output_blocks.%d" % i for i in range(0, 9)]],
                         "rim + CN outputs + UNet outputs at 1280 channels exact": [("controlnet.cond_embed", None), ("controlnet.stem", None), ("controlnet.emb", None), ("head", None), ("stem", None), ("emb", None), ("controlnet", OUT)]
                         + [(z, OUT) for z in ["input_blocks.%d" % i for i in range(7, 12)] + ["middle"] + ["output_blocks.%d" % i for i in range(0, 6)]]}.items():
            rules[:] = rl
            m2, x2 = err()
            print(f"  {name:70s}: mean {m2:.3e} max {x2:.3e}   share of mean^2 {1 - (m2 / m) ** 2:6.1%}", flush=True)
    elif a.mode == "combos":
        OUT = {"lin_out", "conv_out", "norm_out", "attn_out"}
        W = {"lin_w", "conv_w"}
        combos = {
            "controlnet: all exact": [("controlnet", None)],
            "controlnet: outputs exact (fp32 stream in the ControlNet)": [("controlnet", OUT)],
            "controlnet: weights exact": [("controlnet", W)],
            "controlnet: outputs + weights exact": [("controlnet", OUT | W)],
            "controlnet.cond_embed: all exact": [("controlnet.cond_embed", None)],
            "controlnet.cond_embed + stem + emb: all exact": [("controlnet.cond_embed", None), ("controlnet.stem", None), ("controlnet.emb", None)],
            "mergers: outputs exact": [("mergers", OUT)],
            "mergers: all exact": [("mergers", None)],
            "head + stem + emb: all exact": [("head", None), ("stem", None), ("emb", None)],
            "controlnet outputs + mergers outputs + head/stem/emb exact": [("controlnet", OUT), ("mergers", OUT), ("head", None), ("stem", None), ("emb", None)],
            "controlnet all + mergers outputs + head/stem/emb exact": [("controlnet", None), ("mergers", OUT), ("head", None), ("stem", None), ("emb", None)],
            "everything: outputs exact; controlnet + head/stem/emb all exact": [("", OUT), ("controlnet", None), ("head", None), ("stem", None), ("emb", None)],
            "everything: outputs exact; controlnet weights + head/stem/emb all exact": [("", OUT), ("controlnet", W), ("head", None), ("stem", None), ("emb", None)],
        }
        for name, rl in combos.items():
            rules[:] = rl
            m, x = err()
            print(f"  {name:78s}: mean {m:.3e} max {x:.3e}   share of mean^2 {1 - (m / base[0]) ** 2:6.1%}", flush=True)
    else:
        outs = [z for z in zones if z.startswith("output_blocks")]
        tail = ["head"] + outs[::-1]
        for k in range(1, len(tail) + 1):
            exact_zones.clear(); exact_zones.update(tail[:k])
            m, x = err()
            print(f"  exact tail {tail[k - 1]:18s} (last {k:2d} zones): mean {m:.3e} max {x:.3e}", flush=True)


if __name__ == "__main__":
    main()
