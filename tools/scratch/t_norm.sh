cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_i8gram.py tests/test_gpu_cabi_sweep.py -x -q -m gpu 2>&1 | tail -15
for v in 1 0; do
PGL_I8_NORM_GEMM=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > gpurun_out/r3_ab_norm_$v.log 2>&1
python - <<P
import json
l=[x for x in open('gpurun_out/r3_ab_norm_$v.log') if x.startswith('{')][-1]
d=json.loads(l)
print('norm_gemm=$v', round(d['ms_per_step'],1), d['log_likelihood_after'], {k:round(v,1) for k,v in d['stages_ms_rank0'].items() if k.startswith('gram')})
P
done
