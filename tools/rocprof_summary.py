#!/usr/bin/env python
"""Turn a rocprofv3 rocpd SQLite database (gpurun_out/...) into the small CSV/markdown summaries kept under profiles/."""
import glob
import sqlite3
import sys


def main(db_glob, out_prefix, note=""):
    db = sorted(glob.glob(db_glob))[0]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_prefix + "_kernel_stats.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats ; durations in ns ; %s\n" % note)
        f.write("name,calls,total_ns,avg_ns,percent\n")
        for r in rows:
            f.write('"%s",%d,%.0f,%.1f,%.4f\n' % (r[0].replace('"', "'"), r[1], r[2] * 1e3 if r[2] < 1e12 else r[2], r[3] * 1e3 if r[2] < 1e12 else r[3], r[4]))
    print("wrote", out_prefix + "_kernel_stats.csv", len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
