"""Stage view of a rocprofv3 --kernel-trace CSV of tools/probe_flipweights.py (or any run of pgl_sweep): for the LAST batch in the trace, the
launches of the initial tableau sweep (pivot_list .. first decide), the proposal windows (first decide .. active_index) and the weight draw
(active_index .. scatter_active): wall time, busy time (sum of kernel durations), idle time between launches, and the kernels by total time.

    python tools/trace_stages.py <kernel_trace.csv> [list]
"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return n


def main(path, listing=False):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1), int(r["Workgroup_Size_X"])))
    rows.sort()
    names = [r[2] for r in rows]

    def last(prefix, before=None):
        idx = [i for i, n in enumerate(names) if n.startswith(prefix) and (before is None or i < before)]
        return idx[-1] if idx else None
    i_sc = last("scatter_active_kernel")
    i_ai = last("active_index_kernel", i_sc)
    i_pl = last("pivot_list_kernel", i_ai)
    i_d0 = next((i for i in range(i_pl, i_ai) if names[i].startswith("decide_kernel")), i_ai)
    for title, a, b in (("flips.init", i_pl, i_d0), ("flips.windows", i_d0, i_ai), ("weights", i_ai, i_sc + 1)):
        seg = rows[a:b]
        if not seg:
            continue
        wall = (max(r[1] for r in seg) - seg[0][0]) / 1e6
        busy = sum(r[1] - r[0] for r in seg) / 1e6
        gaps = [max(0, seg[i + 1][0] - max(r[1] for r in seg[:i + 1][-4:])) for i in range(len(seg) - 1)]
        agg = defaultdict(lambda: [0, 0.0])
        for s, e, n, g, w in seg:
            agg[n][0] += 1
            agg[n][1] += (e - s) / 1e6
        print("== %s: %d launches, %.2f ms wall, %.2f ms busy, %.2f ms idle between launches (median gap %.1f us)"
              % (title, len(seg), wall, busy, sum(gaps) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0))
        for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("   %9.3f ms %5d  (%8.1f us each)  %s" % (ms, c, ms / c * 1e3, n[:90]))
        if listing:
            t0 = seg[0][0]
            for s, e, n, g, w in seg:
                print("      %10.3f %9.3f  %-60s wgs %d x %d" % ((s - t0) / 1e6, (e - s) / 1e6, n[:60], g // max(w, 1), w))


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2)
