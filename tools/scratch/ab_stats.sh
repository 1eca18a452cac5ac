cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
PGL_I8_STATS_SPLIT=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp64-compare --no-scaling-proxy > gpurun_out/r3_ab_stats_$v.log 2>&1
python - <<P
import json
l=[x for x in open('gpurun_out/r3_ab_stats_$v.log') if x.startswith('{')][-1]
d=json.loads(l)
print('split=$v', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['stages_ms_rank0'].items() if k.startswith('gram')})
P
done
