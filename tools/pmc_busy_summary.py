#!/usr/bin/env python
"""Turn the text summary of tools/pmc_i8_busy.sh (gpurun_out/<tag>_i8busy.txt: per counter the mean value and mean duration over the launches of one
pass) into profiles/i8_busy_pmc.json, the file bench.py quotes as roofline.issue_counters -- with the source hash of the kernel sources the passes
ran on, so that the quote disappears when the kernels change.

    python tools/pmc_busy_summary.py gpurun_out/r05_i8busy.txt profiles/i8_busy_pmc.json
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyglm_amd._lib import source_hash      # noqa: E402


def main(txt, out):
    val, dur = {}, {}
    for ln in open(txt):
        m = re.match(r"(\w+): (\d+) launches, mean value ([0-9.eE+-]+), mean duration ([0-9.]+) ms", ln.strip())
        if m:
            val[m.group(1)], dur[m.group(1)] = float(m.group(3)), float(m.group(4))
    need = ["GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_I8", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]
    missing = [k for k in need if k not in val]
    if missing:
        raise SystemExit("missing counters in %s: %s" % (txt, missing))
    ms = dur["GRBM_GUI_ACTIVE"]
    clock = val["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e-3) * 1e-9                 # GRBM_GUI_ACTIVE is summed over the 8 XCCs
    ms_m = dur["SQ_INSTS_VALU_MFMA_I8"]
    busy = 16.0 * val["SQ_INSTS_VALU_MFMA_I8"] / (1024.0 * clock * 1e9 * ms_m * 1e-3)      # 16 cycles per v_mfma_i32_16x16x64_i8, 1024 SIMDs
    rec = {"cmd": "tools/pmc_i8_busy.sh (separate rocprofv3 --pmc passes, --kernel-trace only, over `python3 tools/probe_i8_real.py 13 4 gram`: one "
                  "group of 8 neurons at the cfg3 shape, 5 launches per pass)",
           "source_hash": source_hash(),
           "kernel": "i8_gram_kernel, 13 planes x 8 neurons per launch",
           "per_launch": {"duration_ms": ms, "GRBM_GUI_ACTIVE_sum_over_8_xcc": val["GRBM_GUI_ACTIVE"], "effective_clock_ghz": round(clock, 3),
                          "SQ_INSTS_VALU_MFMA_I8": val["SQ_INSTS_VALU_MFMA_I8"], "mfma_pipe_busy": round(busy, 3),
                          "SQ_WAIT_ANY_over_SQ_WAVE_CYCLES": round(val["SQ_WAIT_ANY"] / val["SQ_WAVE_CYCLES"], 3)},
           "raw_counters": val, "raw_durations_ms": dur,
           "note": "effective clock = GRBM_GUI_ACTIVE / 8 / duration; MFMA pipe busy = 16 cycles x MFMA instructions / (1024 SIMDs x effective clock x "
                   "duration). frac = achieved / nominal peak ~ busy x effective clock / 2.4 GHz: the kernel runs at the package power limit"}
    for a, b in (("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), ("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")):
        if a in val and b in val:
            rec["per_launch"]["%s_over_%s" % (a, b)] = round(val[a] / val[b], 3)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec["per_launch"], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
