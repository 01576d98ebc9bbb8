#!/bin/bash
# Same-box A/B of two builds of libannsolo_mi.so: scripts/ab_lib.sh <libA.so> <libB.so> [reps]
cd "$(dirname "$0")/.."
reps=${3:-3}
for r in $(seq $reps); do
for lib in "$1" "$2"; do
ASL_LIB_PATH=$(readlink -f $lib) python bench.py --cpu-seconds 0 --recall-queries 0 --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('$lib', 'step', d['ms_per_step'], 'scan', s['scan'], 'rescore', s['rescore'], 'gemm', s['coarse_gemm'])"
done
done
