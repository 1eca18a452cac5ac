#!/bin/bash
# Profiles of the headline command on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      e.g. r04
# -> gpurun_out/<tag>_prof/   rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1` (cfg3)
#    gpurun_out/<tag>_pmc_*/  separate --pmc passes over the integer product kernel (FETCH_SIZE, WRITE_SIZE, TCC hits / misses), one sweep each
# Counters are collected in their own runs with --kernel-trace only (no sys / hip / memory-copy traces), the program itself after `--`.
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-fp64-compare --no-scaling-proxy --no-fixed-state"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$R/gpurun_out/${tag}_pmc_$(echo $C | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex i8_gram_kernel --output-format csv -d $d -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS > $d.log 2>&1
done
d=$R/gpurun_out/${tag}_prof
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $d.log 2>&1
ls -R $R/gpurun_out/${tag}_prof | head -20
