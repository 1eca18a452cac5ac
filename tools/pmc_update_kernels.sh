cd /tmp && export TMPDIR=/tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/r4_pmc_upd_$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -- $GRAFT_REPO_ROOT/tools/bin/ubench_update 256 > /dev/null 2>&1
  f=$(ls $d/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'P'
import csv,sys
from collections import defaultdict
agg=defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if "gemm_tn_f64<2, 2, 1" in n or "update_kernel" in n:
        agg[("generic" if "gemm_tn" in n else "pipeline", int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: "%.4g" % (sum(v)/len(v)) for c,v in agg[k].items()}, len(list(agg[k].values())[0]))
P
done
