"""Static audit of the GEMM kernels' gfx950 ISA (no GPU needed):  python tools/check_isa.py file.s [--dump N]

For every kernel in a `hipcc -S --cuda-device-only` listing: registers / scratch, and for every innermost loop that contains MFMAs (the K loops)
the instruction mix -- MFMA, ds_read, LDS-DMA, other VALU, SALU, branches, waits, scratch traffic.  The loop the product spends its time in
should show: no scratch_*, no branch besides the back edge, and only the DMA instructions' own s_mov m0 / address adds beside the MFMAs.
Also asserts that M0 is written only by the LDS-DMA sequences (glds16_asm does not restore it; svd_common.h)."""
import re
import sys


def kernels(lines):
    cur, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if cur:
                yield cur, lines[start:i]
            cur, start = m.group(1), i
    if cur:
        yield cur, lines[start:]


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_"): return "ds_other"
    if op.startswith("global_load_lds"): return "lds_dma"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"


def loops(body):
    """innermost loops = [label .. last backward branch to label] that contain no other such loop start"""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    spans = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            spans.append((labels[m.group(1)], i))
    # merge spans with the same header (keep the furthest back edge)
    best = {}
    for a, b in spans:
        best[a] = max(best.get(a, b), b)
    spans = sorted(best.items())
    inner = [(a, b) for a, b in spans if not any(a < c and d <= b and (c, d) != (a, b) for c, d in spans)]
    return inner


def main():
    path = sys.argv[1]
    dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else -1
    lines = open(path).read().splitlines()
    bad = 0
    for name, body in kernels(lines):
        meta = {k: None for k in ("NumVgprs", "NumAgprs", "TotalNumSgprs", "ScratchSize", "Occupancy")}
        for l in body:
            m = re.match(r"^; (\w+): (\d+)", l)
            if m and m.group(1) in meta and meta[m.group(1)] is None:
                meta[m.group(1)] = int(m.group(2))
        short = re.sub(r"^_ZN15svd_gemm_detail12_GLOBAL__N_111gemm_kernelINS_7GemmCfgI", "gemm<", name)[:90]
        print(f"== {short}\n   {meta}")
        m0w = [l for l in body if re.search(r"\bm0\b", l) and not re.match(r"^\s+s_mov_b32 m0, s\d+", l) and "global_load_lds" not in l and not l.strip().startswith(";")]
        if m0w:
            bad += 1
            print("   !! M0 touched outside the LDS-DMA sequence:", m0w[:4])
        n = 0
        for a, b in loops(body):
            ops = [l.split()[0] for l in body[a:b + 1] if re.match(r"^\s+[a-z]", l)]
            mix = {}
            for o in ops:
                c = classify(o)
                mix[c] = mix.get(c, 0) + 1
            if mix.get("mfma", 0) == 0:
                continue
            print(f"   loop#{n} lines {a}-{b}: {len(ops)} instr  " + "  ".join(f"{k}={v}" for k, v in sorted(mix.items())))
            if n == dump:
                print("\n".join(body[a:b + 1]))
            n += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
