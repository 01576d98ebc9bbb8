#!/bin/bash
# round 6, first GPU call: new parity tests + the PQ scan beyond the Infinity Cache (baseline + variants)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_distributed.py -x -q -m gpu -k "grouped or plain_bench" > gpurun_out/r06a/pytest.log 2>&1
tail -3 gpurun_out/r06a/pytest.log
timeout 900 python3 scripts/beyond_llc.py --chunks 16 --tag base > gpurun_out/r06a/llc_base.jsonl 2> gpurun_out/r06a/llc_base.err
cat gpurun_out/r06a/llc_base.jsonl | cut -c1-400
for v in V3_DEPTH_2 V3_NT_1 V3_DEPTH_2_V3_NT_1 V3_DEPTH_3 W4_D3; do
  ASL_LIB_PATH=$(readlink -f scripts/tmp/lib_$v.so) timeout 900 python3 scripts/beyond_llc.py --chunks 16 --tag $v > gpurun_out/r06a/llc_$v.jsonl 2> gpurun_out/r06a/llc_$v.err
  python3 - gpurun_out/r06a/llc_$v.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d['tag'], d['chunks'], d['scan_ms'], d['achieved_gbs_36B'], d['frac_of_8tbs'])
PY
done
