#!/bin/bash
# needs the instrumented library: make -C ann_solo_amd/csrc clean all EXTRA=-DASL_ENABLE_DBG
# VALU/SALU instruction counts of the rescoring kernel under its measurement knobs.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pmc_rs
  ASL_RESCORE_DBG=$v rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-include-regex "rescore_score_v2" \
     --output-format csv -d /tmp/pmc_rs -o x -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --recall-queries 0 > /tmp/pmc_rs.log 2>&1
  python3 - "$v" <<'PY'
import csv, sys, collections, glob
f = glob.glob('/tmp/pmc_rs/**/x_counter_collection.csv', recursive=True)
acc = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(f[0])):
    acc[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
n = len(disp)
print('dbg', sys.argv[1], {k: round(v / n / 1e6, 1) for k, v in sorted(acc.items())}, 'M per dispatch')
PY
done
