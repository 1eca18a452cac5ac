#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes of the integer Gram kernel (one directory per counter) into the small files kept under profiles/:
per-launch CSVs and the `int8` block of profiles/gram_pmc.json that bench.py quotes as roofline.traffic.

    python tools/pmc_summary.py <dir with pmc_counter_collection.csv for FETCH_SIZE> <same for WRITE_SIZE> <out prefix> <planes> <group> [<dir for TCC_HIT_sum TCC_MISS_sum>]
FETCH_SIZE is in KB at the L2's fabric side and is doubled for 16 B/lane requests on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import sys


def rows(d, counter, regex="i8_gram_kernel"):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    out = []
    for r in csv.DictReader(open(f)):
        if regex in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out.append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main(dfetch, dwrite, prefix, planes, group, cmd, dhit=None):
    planes, group = int(planes), int(group)
    res = {}
    for name, d in (("FETCH_SIZE", dfetch), ("WRITE_SIZE", dwrite)):
        rs = rows(d, name)
        with open("%s_pmc_i8gram_%s.csv" % (prefix, name), "w") as f:
            f.write("# rocprofv3 --pmc %s --kernel-trace --kernel-include-regex i8_gram_kernel -- %s; one row per launch\n" % (name, cmd))
            f.write("dispatch_id,grid_size,counter,value_KB,duration_ns\n")
            for r in rs:
                f.write("%d,%d,%s,%f,%d\n" % (r[0], r[1], name, r[2], r[3]))
        full = [r for r in rs if r[3] > 0.5 * max(x[3] for x in rs)]          # launches of full groups only
        res[name] = (sum(r[2] for r in full) / len(full), len(full), sum(r[3] for r in full) / len(full) * 1e-9)
    T, D = 100000, 5120
    Dq, Kp = (D + 319) // 320 * 320, (T + 63) // 64 * 64
    fetch = 2.0 * res["FETCH_SIZE"][0] * 1024.0
    write = res["WRITE_SIZE"][0] * 1024.0
    alg = planes * Dq * Kp * (1 + group) + planes * group * Dq * (Dq + 320) // 2
    tiles = (Dq // 320) * (Dq // 320 + 1) // 2
    l2lds = planes * group * tiles * 640 * Kp
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyglm_amd._lib import source_hash
    blk = {"planes": planes, "group": group, "source_hash": source_hash(),
           "FETCH_SIZE_raw_KB_per_launch": res["FETCH_SIZE"][0], "FETCH_SIZE_launches_averaged": res["FETCH_SIZE"][1], "FETCH_SIZE_avg_launch_s": res["FETCH_SIZE"][2],
           "WRITE_SIZE_raw_KB_per_launch": res["WRITE_SIZE"][0], "WRITE_SIZE_launches_averaged": res["WRITE_SIZE"][1], "WRITE_SIZE_avg_launch_s": res["WRITE_SIZE"][2],
           "cmd": "rocprofv3 --pmc <C> --kernel-trace --kernel-include-regex i8_gram_kernel --output-format csv -- %s (separate passes for FETCH_SIZE and WRITE_SIZE)" % cmd,
           "kernel": "i8_gram_kernel (320 x 320 tiles, 3 stages, column-block tile order; v_mfma_i32_16x16x64_i8), cfg3, %d residue planes of %d neurons per launch (launches of full groups only)" % (planes, group),
           "fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
           "algorithmic_bytes_per_launch": alg, "l2_to_lds_bytes_per_launch": l2lds, "l2_hit_rate": 1.0 - fetch / l2lds,
           "note": "FETCH_SIZE is KB at the L2's fabric side (Infinity-Cache hits included) and is doubled per MI355X_MICROARCH.md (16 B/lane requests "
                   "report 1/2 on gfx950). Algorithmic = every plane of X and of the group's omega X read once + the residue tiles written."}
    if dhit:
        hit, miss = rows(dhit, "TCC_HIT_sum"), rows(dhit, "TCC_MISS_sum")
        fullh = [r for r in hit if r[3] > 0.5 * max(x[3] for x in hit)]
        fullm = [r for r in miss if r[3] > 0.5 * max(x[3] for x in miss)]
        h, m = sum(r[2] for r in fullh) / len(fullh), sum(r[2] for r in fullm) / len(fullm)
        blk.update({"TCC_HIT_sum_per_launch": h, "TCC_MISS_sum_per_launch": m, "l2_hit_rate_from_tcc_counters": h / (h + m),
                    "l2_hit_rate_note": "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) over the launch (all L2 requests, reads and the residue stores); "
                                        "l2_hit_rate above is 1 - fetched bytes / bytes staged into LDS"})
    path = os.path.join(os.path.dirname(prefix) or ".", "gram_pmc.json")
    allj = json.load(open(path)) if os.path.exists(path) else {}
    allj["int8_round1"] = allj.get("int8_round1", allj.get("int8"))
    if "int8" in allj and allj["int8"].get("fetch_bytes_corrected") != blk["fetch_bytes_corrected"]:
        allj["int8_round3"] = allj.get("int8_round3", allj["int8"])
    allj["int8"] = blk
    json.dump(allj, open(path, "w"), indent=1)
    print(json.dumps(blk, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6], cmd="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy --no-fixed-state --no-hbm-probe --no-box-ubench", dhit=sys.argv[6] if len(sys.argv) > 6 else None)
