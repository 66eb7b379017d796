import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else "gemm_kernel"
rows = con.execute("select k.name, p.counter_name, p.value from pmc_events p join kernels k on p.event_id = k.id" ).fetchall() if False else None
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
print(cols)
q = cur.execute("select * from counters_collection").fetchall()
agg = collections.defaultdict(list)
ni = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"); ci = cols.index("counter_name"); vi = cols.index("value")
for r in q:
    if pat in r[ni]: agg[r[ci]].append(r[vi])
for k, v in sorted(agg.items()): print(f"{k:28s} n={len(v)} last={v[-1]:.4g} mean={sum(v)/len(v):.4g}")
