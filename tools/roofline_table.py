#!/usr/bin/env python
"""Per-kernel roofline table from rocprofv3 + the package's own work log -- everything needed to recompute `roofline.frac` from profiles/.

    cd /tmp && export TMPDIR=/tmp
    SVD_WORKLOG=$GRAFT_REPO_ROOT/gpurun_out/wl.json rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rt -o rt -- \
        python $GRAFT_REPO_ROOT/bench.py --workload ar_chunk --steps 2 --warmup 1 --no-trace --no-cpu-baseline
    python tools/roofline_table.py gpurun_out/prof_rt/*/rt_results.db gpurun_out/wl.json profiles/r03_roofline_table.txt

rocprofv3 gives the device time of every dispatch by KERNEL NAME; streamingt2v_amd.ops (SVD_WORKLOG) gives, for the same process, the
algorithmic FLOP and HBM bytes of every launch by kernel key (GEMM: tile configuration x A view x element type = one template
instantiation).  The two cover exactly the same dispatches, so work / time per kernel name is exact.  Peaks:
/opt/skills/guides/MI355X_MICROARCH.md -- 2.5 PFLOP/s dense 16-bit MFMA, 8.0 TB/s HBM3E.
"""
import json
import os
import re
import sqlite3
import sys

MFMA_PEAK, HBM_PEAK = 2500.0, 8.0          # TFLOP/s, TB/s
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gemm_configs():
    """cfg id -> (BM, BN, WM, WN, BK, NSTAGE) from csrc/gemm_cfg.h's X(...) table."""
    txt = open(os.path.join(ROOT, "streamingt2v_amd", "csrc", "gemm_cfg.h")).read()
    out = {}
    for m in re.finditer(r"X\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(true|false),\s*(true|false),\s*(\d+)\)", txt):
        i, bm, bn, wm, wn, bk, _, tr, ns = m.groups()
        out[int(i)] = (int(bm), int(bn), int(wm), int(wn), int(bk), int(ns) % 10, tr == "true")
    return out


def gemm_key_of_kernel_name(name, cfgs):
    m = re.search(r"GemmCfg<(\d+), (\d+), (\d+), (\d+), (\d+), (?:true|false), (true|false), (\d+), (?:true|false), (?:true|false)>, (\d+), (Elem\w+)", name)
    if not m:
        return None
    bm, bn, wm, wn, bk, tr, ns, amode, elem = m.groups()
    want = (int(bm), int(bn), int(wm), int(wn), int(bk), int(ns), tr == "true")
    for i, c in cfgs.items():
        if c == want:
            return f"gemm|{i}|{amode}|{'f16' if elem == 'ElemF16' else 'bf16'}"
    return None


def base_name(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.split(r"[<(]", n, 1)[0]


def main(db, wl_path, out=None):
    cfgs = gemm_configs()
    wl = json.load(open(wl_path))
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration) from kernels group by name").fetchall()
    tot = sum(r[2] for r in rows)
    agg = {}                                   # key -> [label, calls, ns]
    for name, calls, ns in rows:
        key = gemm_key_of_kernel_name(name, cfgs)
        if key is None:
            b = base_name(name)
            key = "attn_temporal" if b.startswith("attn_temporal") else b
            label = b
            if b in ("layernorm_kernel", "layernorm_packed_kernel"):          # one work-log entry (ops.layernorm) covers both forms of the kernel
                key, label = "layernorm_kernel", "layernorm_kernel + layernorm_packed_kernel"
            if b in ("ff_geglu_fused_kernel", "ff_geglu_fused8_kernel"):     # one work-log entry (ops.ff_geglu_fused) covers the four- and the eight-wave form
                key, label = "ff_geglu_fused_kernel", "ff_geglu_fused_kernel / ff_geglu_fused8_kernel"
        else:
            _, i, amode, el = key.split("|")
            c = cfgs[int(i)]
            label = f"gemm_kernel<{c[0]}x{c[1]} w{c[2]}x{c[3]} BK{c[4]} S{c[5]}{' T' if c[6] else ''}, view {amode}, {el}> (cfg {i})"
        a = agg.setdefault(key, [label, 0, 0])
        a[1] += calls; a[2] += ns
    lines = [f"# per-kernel roofline table: rocprofv3 kernel trace {os.path.basename(db)} x work log {os.path.basename(wl_path)} (same process)",
             f"# {sum(a[1] for a in agg.values())} dispatches, {tot / 1e6:.1f} ms summed device time; peaks: MFMA {MFMA_PEAK:.0f} TFLOP/s dense 16 bit, HBM {HBM_PEAK:.1f} TB/s",
             f"# TFLOP/s and TB/s are ALGORITHMIC work (FLOP = 2MNK / 4 N_q N_k d per head; bytes = every operand moved once) / measured device time",
             f"{'kernel':86s} {'calls':>7s} {'wl_calls':>8s} {'total_ms':>9s} {'avg_us':>9s} {'%time':>6s} {'TFLOP/s':>8s} {'mfma_frac':>9s} {'TB/s':>6s} {'hbm_frac':>8s} bound"]
    fam = {"gemm": [0.0, 0.0, 0], "all": [0.0, 0.0]}
    for key, (label, calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        w = wl.get(key)
        tf = tb = None
        if w:
            tf = w[1] / (ns * 1e-9) / 1e12 if w[1] else None
            tb = w[2] / (ns * 1e-9) / 1e12 if w[2] else None
            if key.startswith("gemm|"):
                fam["gemm"][0] += w[1]; fam["gemm"][1] += ns; fam["gemm"][2] += calls
        bound = "-"
        if tf is not None or tb is not None:
            bound = "mfma" if (tf or 0) / MFMA_PEAK >= (tb or 0) / HBM_PEAK else "hbm"
        f = lambda v, fmt: (fmt % v) if v is not None else "-"
        lines.append(f"{label[:86]:86s} {calls:7d} {(w[0] if w else 0):8d} {ns / 1e6:9.2f} {ns / calls / 1e3:9.2f} {100 * ns / tot:6.2f} "
                     f"{f(tf, '%8.1f'):>8s} {f(tf / MFMA_PEAK if tf else None, '%9.3f'):>9s} {f(tb, '%6.2f'):>6s} {f(tb / HBM_PEAK if tb else None, '%8.3f'):>8s} {bound}")
    if fam["gemm"][1]:
        g = fam["gemm"]
        lines.insert(3, f"# GEMM family (all instantiations): {g[2]} dispatches, {g[1] / 1e6:.1f} ms = {100 * g[1] / tot:.1f} % of device time, "
                        f"{g[0] / (g[1] * 1e-9) / 1e12:.1f} TFLOP/s = {g[0] / (g[1] * 1e-9) / 1e12 / MFMA_PEAK:.3f} of the MFMA peak")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
