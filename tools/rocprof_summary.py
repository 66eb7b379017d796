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
    # GPU idle inside the steady state (last 40 % of the trace): union of the kernel intervals against the span, and the gaps between them
    t_lo = span[0] + int(0.6 * (span[1] - span[0]))
    iv = con.execute("select start, end from kernels where start >= ? order by start", (t_lo,)).fetchall()
    if len(iv) > 100:
        busy, gaps, cur_end = 0, [], iv[0][0]
        for st, en in iv:
            if st > cur_end:
                gaps.append(st - cur_end)
            busy += max(0, en - max(st, cur_end))
            cur_end = max(cur_end, en)
        sp = cur_end - iv[0][0]
        big = [g for g in gaps if g > 20000]
        lines.insert(2, f"# steady state (last 40 % of the trace, {len(iv)} dispatches): GPU busy {100.0 * busy / sp:.1f} % of {sp / 1e6:.1f} ms; "
                        f"{len(gaps)} gaps, mean {sum(gaps) / max(len(gaps), 1) / 1e3:.1f} us, {len(big)} gaps > 20 us totalling {sum(big) / 1e6:.1f} ms")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
