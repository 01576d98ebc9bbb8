cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r03 > /dev/null 2>&1
bash scripts/profile_round.sh r03_ivfflat_np112 --index ivfflat --nprobe 112 > /dev/null 2>&1
python bench.py --workload cascade > gpurun_out/r03_cascade_bench.json 2> /dev/null
python bench.py --workload cascade --index ivfflat --nprobe 112 > gpurun_out/r03_cascade_ivfflat_np112_bench.json 2>/dev/null
python bench.py --no-pipeline --cpu-seconds 0 --no-fixed-recall --recall-queries 0 > gpurun_out/r03_serial_bench.json 2>/dev/null
for idx in ivfpq ivfflat; do for W in 2 4 8; do python scripts/sim_rank.py $W 2100000 16384 0 $idx 2>/dev/null | tail -1; done; done > gpurun_out/r03_sim_rank_final.txt
python -c "
import json
for f in ('r03/bench.json','r03_ivfflat_np112/bench.json','r03_cascade_bench.json','r03_cascade_ivfflat_np112_bench.json','r03_serial_bench.json'):
    d=json.load(open('gpurun_out/'+f)); print(f, d['value'], d['ms_per_step'], d.get('stages_ms_per_step'))
d=json.load(open('gpurun_out/r03/bench.json')); print(d['fixed_recall']['value'], d['fixed_recall']['ms_per_step'], d['roofline']['frac'])
"
cat gpurun_out/r03_sim_rank_final.txt
