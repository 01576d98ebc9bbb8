#!/bin/bash
cd "$(dirname "$0")/.."
for d in 0 2 6 3; do
echo -n "dbg=$d "
ASL_FI_DBG=$d timeout 120 python bench.py --index ivfflat --cpu-seconds 0 --recall-queries 0 --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('scan', s['scan'])"
done
