#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (SQLite) kernel trace into a per-kernel stats table
(name, calls, total ms, avg us, min us, max us, % of GPU kernel time) — the same content as
rocprofv3's kernel_stats.csv, written as plain text for profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    stats = {}
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    for n, s, e in rows:
        d = (e - s) / 1e3
        st = stats.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    lines = [f"# {db_path}: {len(rows)} kernel dispatches, {total / 1e3:.3f} ms total kernel time, "
             f"{(t1 - t0) / 1e6:.3f} ms first-start to last-end",
             f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel"]
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{v[0]:7d} {v[1] / 1e3:10.3f} {v[1] / v[0]:10.1f} {v[2]:10.1f} {v[3]:10.1f} "
                     f"{100 * v[1] / total:6.2f}  {k}")
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
