cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err
tail -c 3000 gpurun_out/r3_bench_a.err | tail -5
python -c "
import json
d=json.load(open('gpurun_out/r3_bench_a.json'))
print(d['value'], d['ms_per_step'], d['stages_ms_per_step'])
print(json.dumps(d['roofline']))
print(json.dumps(d['fixed_recall'], indent=0))
print(json.dumps(d['cpu_baseline']))
"
