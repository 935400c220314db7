#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the --stats style table: per kernel calls / total / avg / min / max.
Usage: python tools/rocpd_summary.py <results.db> [> profiles/<name>_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
    q = f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    print(f"{'KERNEL':<110} {'CALLS':>7} {'TOTAL_ms':>10} {'AVG_us':>10} {'MIN_us':>10} {'MAX_us':>10} {'%':>6}")
    for n, c, t, mn, mx in rows:
        n = n if len(n) <= 108 else n[:105] + "..."
        print(f"{n:<110} {c:>7} {t / 1e6:>10.3f} {t / c / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100 * t / total:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
