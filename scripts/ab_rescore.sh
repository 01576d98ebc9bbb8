#!/bin/bash
# A/B of the rescoring kernel: ASL_RESCORE_DBG bits 1 = no probing/resolve, 4 = no peak loads,
# 8 = one candidate per wave (no pairing)
cd "$(dirname "$0")/.."
for v in "$@"; do
ASL_RESCORE_DBG=$v python bench.py --cpu-seconds 0 --recall-queries 0 --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg', $v, 'rescore ms', d['stages_ms_per_step']['rescore'], 'step', d['ms_per_step'])"
done
