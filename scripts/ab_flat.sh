#!/bin/bash
# Same-box A/B of builds on the IVF-Flat step: scripts/ab_flat.sh lib1.so lib2.so ...
cd "$(dirname "$0")/.."
for lib in "$@"; do
ASL_LIB_PATH=$(readlink -f $lib) python bench.py --index ivfflat --cpu-seconds 0 --recall-queries 0 --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('$lib', 'step', d['ms_per_step'], 'scan', s['scan'], 'rescore', s['rescore'])"
done
