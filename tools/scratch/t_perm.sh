cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_perm_prof -- python3 $GRAFT_REPO_ROOT/tools/probe_flipweights.py 0.5 0.5 2 > $GRAFT_REPO_ROOT/gpurun_out/r3_perm_prof.log 2>&1 )
grep -v "^W2026\|^E2026" gpurun_out/r3_perm_prof.log | tail -3
rm -f gpurun_out/r3_perm_prof/*/*kernel_trace.csv
grep -h "permute_tableau\|gather_active" gpurun_out/r3_perm_prof/*/*kernel_stats.csv | cut -c1-200
