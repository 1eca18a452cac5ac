cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r3_final_tests.log 2>&1; tail -3 gpurun_out/r3_final_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_final_bench_cfg3.log 2>&1; tail -c 200 gpurun_out/r3_final_bench_cfg3.log; echo
for c in cfg1 cfg2 cfg4; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_final_bench_$c.log 2>&1; done
python bench.py --config cfg5 --steps 1 --warmup 1 > gpurun_out/r3_final_bench_cfg5.log 2>&1
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_final_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > $GRAFT_REPO_ROOT/gpurun_out/r3_final_prof.log 2>&1 )
for C in FETCH_SIZE WRITE_SIZE; do ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --kernel-include-regex i8_gram_kernel --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_final_pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > $GRAFT_REPO_ROOT/gpurun_out/r3_final_pmc_$C.log 2>&1 ); done
ls gpurun_out/r3_final_prof/*/ gpurun_out/r3_final_pmc_FETCH_SIZE/*/ 2>&1 | head
# keep the merged output small: drop the per-launch kernel trace of the stats run
rm -f gpurun_out/r3_final_prof/*/*kernel_trace.csv
