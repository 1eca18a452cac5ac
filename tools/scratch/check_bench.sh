cd $GRAFT_REPO_ROOT
python bench.py --config cfg1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_chk_cfg1.log 2>&1; tail -c 300 gpurun_out/r3_chk_cfg1.log; echo
python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_chk_cfg2.log 2>&1; tail -c 300 gpurun_out/r3_chk_cfg2.log; echo
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > gpurun_out/r3_chk_cfg3.log 2>&1; tail -c 300 gpurun_out/r3_chk_cfg3.log; echo
