"""profiles/*traffic_signatures*.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/gemm_sig_run.py.
Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): counters in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
coalesced streaming reads -> reads are doubled; WRITE_SIZE as is.
    python tools/pmc_signature.py <signature> <fetch.db> <write.db> <out.json>"""
import json, os, sqlite3, sys


def total(db, counter):
    rows = sqlite3.connect(db).execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    rows = [(n, v) for n, v in rows if "gemm_kernel" in n]
    return len(rows), sum(v for _, v in rows)


def main(sig, fetch_db, write_db, out):
    nf, f = total(fetch_db, "FETCH_SIZE")
    nw, w = total(write_db, "WRITE_SIZE")
    n = max(nf, nw)
    rd, wr = 2.0 * f * 1024.0 / n, w * 1024.0 / n
    doc = json.load(open(out)) if os.path.exists(out) else {"note": "HBM bytes per launch of single GEMM signatures launched alone under rocprofv3 --pmc "
                                                            "(FETCH_SIZE x2 gfx950 correction, WRITE_SIZE as is; KiB -> bytes)", "signatures": {}}
    doc["signatures"][sig] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr}
    json.dump(doc, open(out, "w"), indent=1)
    print(sig, f"read {rd / 1e6:.1f} MB + write {wr / 1e6:.1f} MB per launch over {n} launches")


if __name__ == "__main__":
    main(*sys.argv[1:5])
