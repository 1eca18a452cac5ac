cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_i8gram.py tests/test_gpu_cabi_sweep.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
for v in 1 0 1 0; do
PGL_I8_CRT_OVERLAP=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > gpurun_out/r3_ab_ovl_$v.log 2>&1
python - <<P
import json
l=[x for x in open('gpurun_out/r3_ab_ovl_$v.log') if x.startswith('{')][-1]
d=json.loads(l)
print('overlap=$v', round(d['ms_per_step'],1), d['log_likelihood_after'], {k:round(v,1) for k,v in d['stages_ms_rank0'].items() if k.startswith('gram')}, d['per_rank'][0]['host_only_ms_per_step'])
P
done
