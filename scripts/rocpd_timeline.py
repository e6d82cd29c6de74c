#!/usr/bin/env python3
"""Timeline of the dispatches whose kernel name matches a pattern, from a rocprofv3 rocpd database:
rocpd_timeline.py <results.db> <substring> [last N]   -> start offset (ms from the first listed dispatch), duration (ms), name"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = cur.execute("select name, start, end from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall()[-n:]
t0 = rows[0][1] if rows else 0
for name, st, en in rows:
    print("%9.3f ms  +%8.3f ms  %s" % ((st - t0) / 1e6, (en - st) / 1e6, name[:90]))
