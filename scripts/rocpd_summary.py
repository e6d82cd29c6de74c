#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs: per-kernel stats and per-kernel PMC counter averages.
usage: rocpd_summary.py <results.db> [...]  -> markdown-ish text on stdout"""
import os
import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path); cur = db.cursor()
    print("## %s" % path)
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    try:
        rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
        tot = sum(r[5] for r in rows) or 1
        print("| kernel | calls | avg_us | min_us | max_us | total_ms | pct |")
        print("|---|---|---|---|---|---|---|")
        for r in rows[:int(os.environ.get("ROCPD_ROWS", "12"))]:
            print("| %s | %d | %.1f | %.1f | %.1f | %.2f | %.1f |" % (r[0][:70], r[1], r[2]/1e3, r[3]/1e3, r[4]/1e3, r[5]/1e6, 100*r[5]/tot))
    except Exception as e:
        print("kernels view failed:", e, cols)
    try:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
        if rows:
            print("\n| kernel | counter | dispatches | avg per dispatch | sum |")
            print("|---|---|---|---|---|")
            for r in rows:
                if any(k in r[0] for k in os.environ.get('ROCPD_KERNELS', 'k_seed,k_bsw,k_gather').split(',')):
                    print("| %s | %s | %d | %.4g | %.4g |" % (r[0][:40], r[1], r[2], r[3], r[4]))
    except Exception as e:
        print("counters view failed:", e)
    print()
