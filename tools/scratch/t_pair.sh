cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi_sweep.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_update.py -x -q -m gpu 2>&1 | tail -3
for a in "0.47 0.45" "0.77 0.9" "0.5 0.5"; do
for v in 1 0; do
echo "pair=$v args $a"; PGL_FLIP_PAIR=$v python tools/probe_flipweights.py $a 2 2>&1 | tail -1 | cut -c1-330
done; done
