#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/*.json.
Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of wide coalesced streaming reads (16 B/lane), so reads are doubled; WRITE_SIZE is taken as is."""
import collections, json, re, sqlite3, sys

def per_kernel(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, v in rows:
        agg[n][0] += 1; agg[n][1] += v
    return agg

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("svd_gemm_detail::", "")
    m = re.search(r"gemm_kernel<GemmCfg<(\d+), (\d+), (\d+), (\d+), (\d+), \w+, (\w+)(?:, [^>]*)?>, (\d), Elem(\w+)>", n)
    if m:
        return f"gemm {m.group(1)}x{m.group(2)} bk{m.group(5)} mode{m.group(7)}{' T' if m.group(6) == 'true' else ''} {m.group(8).lower()}"
    return re.sub(r"^void ", "", n).split("(")[0]

def main(fetch_db, write_db, out):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for n in set(f) | set(w):
        cnt = max(f.get(n, [0, 0])[0], w.get(n, [0, 0])[0])
        rd = 2.0 * f.get(n, [0, 0.0])[1] * 1024.0          # gfx950 correction x2
        wr = w.get(n, [0, 0.0])[1] * 1024.0
        k = short(n)
        a = res.setdefault(k, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
        a["launches"] += cnt; a["read_bytes"] += rd; a["write_bytes"] += wr
    for k, a in res.items():
        a["bytes_per_launch"] = (a["read_bytes"] + a["write_bytes"]) / max(a["launches"], 1)
    json.dump({"note": "HBM traffic per kernel (FETCH_SIZE x2 correction on gfx950, WRITE_SIZE as is; KiB -> bytes)",
               "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["read_bytes"] - kv[1]["write_bytes"]))}, open(out, "w"), indent=1)
    for k, a in list(sorted(res.items(), key=lambda kv: -kv[1]["read_bytes"] - kv[1]["write_bytes"]))[:14]:
        print(f"{k:44s} x{a['launches']:5d}  read {a['read_bytes']/1e9:8.2f} GB  write {a['write_bytes']/1e9:8.2f} GB  per launch {a['bytes_per_launch']/1e6:9.1f} MB")

if __name__ == "__main__":
    main(*sys.argv[1:4])
