"""Per-launch view of a rocprofv3 --kernel-trace CSV of a bench.py run: the launches of the LAST sweep's first batch between the Gram of
one batch and the next, in order, with durations (ms), so that the flip / weight stages can be read launch by launch."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n


def main(path, which=-1):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
    rows.sort()
    # a batch's flip/weight phase = launches after the last i8_crt_kernel of a run of groups until the next i8_colstats
    crt = [i for i, r in enumerate(rows) if r[2].startswith("i8_crt_kernel")]
    ends = [i for k, i in enumerate(crt) if k + 1 == len(crt) or not any(rows[j][2].startswith("i8_gram") for j in range(i + 1, min(len(rows), i + 6)))]
    i0 = ends[which]
    j = i0 + 1
    out = []
    while j < len(rows) and not rows[j][2].startswith("i8_colstats") and not rows[j][2].startswith("pack_weights"):
        out.append(rows[j])
        j += 1
    t0 = out[0][0]
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, n, g, w in out:
        agg[n][0] += 1
        agg[n][1] += (e - s) / 1e6
    print("phase: %d launches, %.1f ms wall, %.1f ms busy" % (len(out), (out[-1][1] - t0) / 1e6, sum((e - s) for s, e, *_ in out) / 1e6))
    for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%9.2f ms %5d  %s" % (ms, c, n))
    if len(sys.argv) > 3:
        for s, e, n, g, w in out:
            print("%10.3f %9.3f  %-50s grid %d x %d" % ((s - t0) / 1e6, (e - s) / 1e6, n[:50], g // max(w, 1), w))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -1)
