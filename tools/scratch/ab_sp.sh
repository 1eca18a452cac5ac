cd $GRAFT_REPO_ROOT
for v in 6 8 6 8; do
PGL_CHOL_SP=$v python bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3_ab_sp_$v.log 2>&1
python - <<P
import json
l=[x for x in open('gpurun_out/r3_ab_sp_$v.log') if x.startswith('{')][-1]
d=json.loads(l)
print('SP=$v', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['stages_ms_rank0'].items() if k in ('weights','gram.int8','activation')})
P
done
