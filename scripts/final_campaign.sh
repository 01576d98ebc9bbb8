#!/bin/bash
# End-of-round campaign on the final build: the four fuzzers (fresh seeds), the whole GPU suite, the bench lines
# that go to profiles/.   scripts/final_campaign.sh <tag> <seed base>
cd "$(dirname "$0")/.."
tag=$1; seed=${2:-660000}
out=gpurun_out/$tag; mkdir -p $out
echo "HEAD $(cat .git_head 2>/dev/null)" > $out/campaign.txt
timeout 700 python3 scripts/fuzz_paths.py 540 $((seed+1)) 2>&1 | grep -v amdgpu.ids | tail -3 > $out/fuzz_paths.log
timeout 500 python3 scripts/fuzz_shards.py 360 $((seed+2)) 2>&1 | grep -v amdgpu.ids | tail -3 > $out/fuzz_shards.log
timeout 300 python3 scripts/fuzz_flat.py 180 $((seed+3)) 2>&1 | grep -v amdgpu.ids | tail -3 > $out/fuzz_flat.log
timeout 300 python3 scripts/fuzz_pre.py 180 $((seed+4)) 2>&1 | grep -v amdgpu.ids | tail -3 > $out/fuzz_pre.log
tail -n 2 $out/fuzz_*.log
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $out/pytest_gpu.log
cat $out/pytest_gpu.log
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --index ivfflat --nprobe 112 > $out/ivfflat_np112_bench.json 2> $out/ivfflat_np112_bench.err
python3 -c "
import json
d=json.load(open('$out/bench.json')); print({k:d[k] for k in ('value','ms_per_step','value_at_fixed_recall','value_at_reference_batch')}, d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['beyond_llc']['frac'])
d=json.load(open('$out/ivfflat_np112_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))
"
