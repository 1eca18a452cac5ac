cd $GRAFT_REPO_ROOT/pyglm_amd/csrc
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPGL_DECIDE_TIMING -c pgl_flips.hip -o /tmp/pgl_flips_t.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpyglm_hip.so _build/pgl_api.o _build/pgl_gemm.o _build/pgl_elementwise.o /tmp/pgl_flips_t.o _build/pgl_chol.o _build/pgl_i8gram.o _build/pgl_sweep.o _build/pgl_update.o
cd $GRAFT_REPO_ROOT
python tools/scratch/decide_timing.py 0.5 0.5 2>&1 | tail -6
python tools/scratch/decide_timing.py 0.77 0.9 2>&1 | tail -6
