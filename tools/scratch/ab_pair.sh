cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
PGL_FLIP_PAIR=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > gpurun_out/r3_ab_pair_$v.log 2>&1
python - <<P
import json
l=[x for x in open('gpurun_out/r3_ab_pair_$v.log') if x.startswith('{')][-1]
d=json.loads(l)
print('pair=$v', round(d['ms_per_step'],1), d['log_likelihood_after'], {k:round(v,1) for k,v in d['stages_ms_rank0'].items() if k.startswith('flips') or k=='weights'})
P
done
