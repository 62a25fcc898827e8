#!/usr/bin/env python
"""Pivot the counters of one `rocprofv3 --pmc <counters> --kernel-trace` pass (rocpd SQLite) into a
per-kernel table: calls, total ms, and the SUM of every collected counter, plus a few derived
columns when their inputs are present (MI355X_MICROARCH.md units: GRBM_GUI_ACTIVE is summed over
the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_*
count quad-cycles per wave).

    python tools/rocpd_sq.py <results.db> [out.txt]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", name)
    return name[:64]


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    dur = {}
    for n, s, e in cur.execute(f"select {name_col}, start, end from kernels"):
        d = dur.setdefault(short(n), [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e6
    agg, names = {}, []
    for n, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if c not in names:
            names.append(c)
        k = agg.setdefault(short(n), {})
        k[c] = k.get(c, 0.0) + v
    lines = [f"# {path}", "kernel".ljust(64) + f"{'calls':>6} {'ms':>9} " + " ".join(f"{c[-18:]:>18}" for c in names) + "  derived"]
    for k, (calls, ms) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:16]:
        c = agg.get(k, {})
        d = []
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui:
            d.append(f"clk={gui / 8 / (ms * 1e6):.2f}GHz")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                d.append(f"mfma_busy={100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8 * 1024):.1f}%")
            if "SQ_ACTIVE_INST_LDS" in c:
                d.append(f"lds_inst_active={100 * 4 * c['SQ_ACTIVE_INST_LDS'] / (gui / 8 * 1024):.1f}%")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for nm in names:
                if nm.startswith(("SQ_WAIT", "SQ_ACTIVE_INST")):
                    d.append(f"{nm[3:].lower()}={100 * c[nm] / wc:.1f}%")
        lines.append(k.ljust(64) + f"{calls:6d} {ms:9.2f} " + " ".join(f"{c.get(n, 0):18.4g}" for n in names) + "  " + " ".join(d))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
