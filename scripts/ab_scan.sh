#!/bin/bash
# needs the instrumented library: make -C ann_solo_amd/csrc clean all EXTRA=-DASL_ENABLE_DBG
# A/B of the PQ scan kernel with measurement knobs (scan-variant = kernel | dbg<<8):
#   dbg bit 1: no top-k appends, 2: no LUT build, 4: no ADC, 8: no code loads
cd "$(dirname "$0")/.."
for v in "$@"; do
python bench.py --cpu-seconds 0 --recall-queries 0 --steps 5 --warmup 1 --scan-variant $v 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant', $v, 'scan ms', d['stages_ms_per_step']['scan'], 'step', d['ms_per_step'])"
done
