cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi_sweep.py tests/test_gpu_update.py -x -q -m gpu 2>&1 | tail -3
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for a in "0.5 0.5"; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_dec_prof -- python3 $GRAFT_REPO_ROOT/tools/probe_flipweights.py $a 2 > $GRAFT_REPO_ROOT/gpurun_out/r3_dec_prof.log 2>&1 )
grep -v "^W2026\|^E2026" gpurun_out/r3_dec_prof.log | tail -2
rm -f gpurun_out/r3_dec_prof/*/*kernel_trace.csv
cut -d, -f1-4 gpurun_out/r3_dec_prof/*/*kernel_stats.csv | head -24 | cut -c1-150
rm -rf gpurun_out/r3_dec_prof
done
