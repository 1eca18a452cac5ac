cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_i8gram.py -x -q -m gpu 2>&1 | tail -3
python bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3_n2_cfg5.log 2>&1
python bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_n2_cfg4.log 2>&1
python - <<P
import json
for c in ('cfg5','cfg4'):
    l=[x for x in open('gpurun_out/r3_n2_%s.log'%c) if x.startswith('{')][-1]
    d=json.loads(l)
    print(c, round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['stages_ms_rank0'].items()})
P
