#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd SQLite .db, or *_kernel_stats.csv) into a small text table
that is committed under profiles/ (the raw .db stays in gpurun_out/ scratch)."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:110]


def main(path, out=None):
    con = sqlite3.connect(path)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    span = con.execute("select min(start), max(end) from kernels").fetchone()
    lines = [f"# rocprofv3 kernel-trace summary of {path}",
             f"# kernels: {sum(r[1] for r in rows)} dispatches, {tot / 1e6:.2f} ms summed device time, "
             f"{(span[1] - span[0]) / 1e6:.2f} ms first-start..last-end",
             f"{'name':112s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"{short(n):112s} {c:6d} {t / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * t / tot:6.2f}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
