#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (*.db) into a per-kernel table: calls, total/avg/min/max duration.
Usage: python scripts/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                       f"group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total_ms | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        lines.append(f"| `{short}` | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / total:.1f} |")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
