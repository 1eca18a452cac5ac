#!/bin/bash
# Everything profiles/ needs at the final sources, in one gpurun call from the repo root:   tools/profile_final.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1` (cfg3)          -> gpurun_out/<tag>_prof/ + <tag>_cfg3_1gpu_kernel_stats.csv
#   2. separate --pmc passes over i8_gram_kernel in one bench sweep (FETCH_SIZE, WRITE_SIZE, TCC hits/misses) -> gram_pmc.json (with source_hash)
#   3. tools/pmc_i8_busy.sh: issue counters of the product kernel                           -> i8_busy_pmc.json (with source_hash)
# Counters are collected in their own runs with --kernel-trace only, the program itself after `--`.
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-fp64-compare --no-scaling-proxy --no-fixed-state --no-hbm-probe --no-box-ubench"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/${tag}_pmc_$(echo $C | tr ' ' '_')
  rm -rf $d
  (cd $R && rocprofv3 --pmc $C --kernel-trace --kernel-include-regex i8_gram_kernel --output-format csv -d $d -- python3 bench.py --steps 1 --warmup 0 $ARGS) > $d.log 2>&1
done
d=$O/${tag}_prof
rm -rf $d
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 $ARGS) > $d.log 2>&1
cd $R
mkdir -p $O/${tag}_keep
f=$(find $O/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/${tag}_keep/${tag}_cfg3_1gpu_kernel_stats.csv
cp profiles/gram_pmc.json $O/${tag}_keep/gram_pmc.json
python3 tools/pmc_summary.py $O/${tag}_pmc_FETCH_SIZE $O/${tag}_pmc_WRITE_SIZE $O/${tag}_keep/${tag}_cfg3 13 8 $O/${tag}_pmc_TCC_HIT_sum_TCC_MISS_sum > $O/${tag}_keep/pmc_summary.txt 2>&1
bash tools/pmc_i8_busy.sh $tag > /dev/null 2>&1
python3 tools/pmc_busy_summary.py $O/${tag}_i8busy.txt $O/${tag}_keep/i8_busy_pmc.json > $O/${tag}_keep/busy_summary.txt 2>&1
# the traces themselves are large: keep the summaries only
rm -rf $O/${tag}_prof $O/${tag}_pmc_FETCH_SIZE $O/${tag}_pmc_WRITE_SIZE $O/${tag}_pmc_TCC_HIT_sum_TCC_MISS_sum $O/${tag}_i8busy_*
ls -la $O/${tag}_keep
