#!/usr/bin/env python
"""Per-call kernel durations, in launch order, from a rocprofv3 rocpd SQLite database: `rocprof_calls.py <db glob> [max rows]`."""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1]))[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("# columns:", cols, file=sys.stderr)
rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
lim = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
t0 = rows[0][1]
for name, s, e, gx, wx in rows[:lim]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%10.3f ms  +%9.3f ms  grid %8d x %4d  %s" % ((s - t0) * 1e-6, (e - s) * 1e-6, gx // max(wx, 1), wx, short[:70]))
