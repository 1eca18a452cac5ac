cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3_final6_tests.log; cat gpurun_out/r3_final6_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_final6_bench_cfg3.log 2>&1; tail -c 150 gpurun_out/r3_final6_bench_cfg3.log; echo
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_final6_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > $GRAFT_REPO_ROOT/gpurun_out/r3_final6_prof.log 2>&1 )
rm -f gpurun_out/r3_final6_prof/*/*kernel_trace.csv
for c in cfg1 cfg2 cfg4; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_final6_bench_$c.log 2>&1; done
python bench.py --config cfg5 --steps 1 --warmup 1 > gpurun_out/r3_final6_bench_cfg5.log 2>&1
