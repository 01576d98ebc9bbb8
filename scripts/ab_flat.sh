#!/bin/bash
# Same-box A/B of builds on the IVF-Flat step: [NPROBE=112] [REPS=2] scripts/ab_flat.sh lib1.so lib2.so ...
cd "$(dirname "$0")/.."
for r in $(seq ${REPS:-2}); do
for lib in "$@"; do
ASL_LIB_PATH=$(readlink -f $lib) python bench.py --index ivfflat --nprobe ${NPROBE:-112} --cpu-seconds 0 --recall-queries 0 --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; r=d['roofline']; print('$lib', 'step', d['ms_per_step'], 'scan', s['scan'], 'rescore', s['rescore'], 'frac', r['frac'], 'frac_by_lines', r.get('frac_by_lines'))"
done
done
