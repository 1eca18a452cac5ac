cd $GRAFT_REPO_ROOT
for g in auto int8 auto int8; do
python bench.py --config cfg2 --steps 10 --warmup 3 --no-cpu-baseline --gram $g > gpurun_out/r3_cfg2_$g.log 2>&1
python - <<P
import json
l=[x for x in open('gpurun_out/r3_cfg2_$g.log') if x.startswith('{')][-1]
d=json.loads(l)
print('$g', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_rank0'].items() if k.startswith('gram')}, d['log_likelihood_after'])
P
done
