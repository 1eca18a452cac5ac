#!/bin/bash
# MFMA-busy, wait and L2 counters of the integer product kernel on the bench's own data (one group of 8 neurons at the cfg3 shape):
#   [PZERO=1] [PGL_PROBE_LIB=ab] tools/pmc_i8_busy.sh <tag> ["counter sets"...]   (through gpurun, from the repo root)
#   -> gpurun_out/<tag>_i8busy_*/ + gpurun_out/<tag>_i8busy.txt
# Separate --pmc passes, --kernel-trace only, the program itself after `--`.
tag=${1:-r04}${PZERO:+_zero}${PGL_PROBE_LIB:+_$PGL_PROBE_LIB}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/${tag}_i8busy.txt
: > $out
if [ $# -eq 0 ]; then
  set -- "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
fi
for C in "$@"; do
  d=$R/gpurun_out/${tag}_i8busy_$(echo $C | tr ' ' '_')
  rm -rf $d
  (cd $R && rocprofv3 --pmc $C --kernel-trace --kernel-include-regex i8_gram_kernel --output-format csv -d $d -- python3 tools/probe_i8_real.py 13 4 gram) > $d.log 2>&1
  python3 - $d >> $out <<'PY'
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
if not fs:
    print("no counter file in", sys.argv[1]); sys.exit(0)
acc = {}
for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
    if "i8_gram_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in acc.items():
    print("%s: %d launches, mean value %.6g, mean duration %.3f ms" % (k, len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v) * 1e-6))
PY
done
cat $out
