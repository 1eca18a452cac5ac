cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_final3_bench_cfg3.log 2>&1; tail -c 150 gpurun_out/r3_final3_bench_cfg3.log; echo
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_final3_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > $GRAFT_REPO_ROOT/gpurun_out/r3_final3_prof.log 2>&1 )
rm -f gpurun_out/r3_final3_prof/*/*kernel_trace.csv
for C in FETCH_SIZE WRITE_SIZE; do ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --kernel-include-regex i8_gram_kernel --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_final3_pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > $GRAFT_REPO_ROOT/gpurun_out/r3_final3_pmc_$C.log 2>&1 ); done
for c in cfg1 cfg2 cfg4; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_final3_bench_$c.log 2>&1; done
python bench.py --config cfg5 --steps 1 --warmup 1 > gpurun_out/r3_final3_bench_cfg5.log 2>&1
