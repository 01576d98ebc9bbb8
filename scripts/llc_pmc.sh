#!/bin/bash
# PQ scan beyond the Infinity Cache: kernel trace + PMC passes (separate runs, --kernel-trace only)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r06b; mkdir -p $out
CH=${1:-16}
timeout 1200 python3 scripts/beyond_llc.py --chunks 64 --tag base > $out/llc_base_64.jsonl 2> $out/llc_base_64.err
cut -c1-330 $out/llc_base_64.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/llc_kt -o x -- python3 scripts/beyond_llc.py --only $CH --chunks $CH --tag kt > $out/llc_under_rocprof.jsonl 2> /tmp/llc_kt.err
f=$(find /tmp/llc_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats.csv
t=$(find /tmp/llc_kt -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 - "$t" > $out/scan_launches.txt <<'PY'
import csv, sys
v = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in csv.DictReader(open(sys.argv[1])) if 'pq_scan_v3_kernel' in r['Kernel_Name']]
print('pq_scan_v3_kernel dispatches under rocprofv3 --kernel-trace (ms):', [round(x, 3) for x in v])
print('the timed ones (last 3): avg %.3f ms' % (sum(v[-3:]) / 3))
PY
cat $out/scan_launches.txt
pmc() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "pq_scan" --output-format csv -d /tmp/llc_$name -o x -- python3 scripts/beyond_llc.py --only $CH --chunks $CH --tag $name > /tmp/llc_$name.log 2>&1
  python3 - "$name" >> $out/pmc_summary.txt <<'PY'
import csv, sys, collections, glob
name = sys.argv[1]
f = glob.glob(f'/tmp/llc_{name}/**/x_counter_collection.csv', recursive=True)
if not f:
    print(name, 'no counter file'); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in acc:
    print(f'[{name}] {k} dispatches={cnt[k]}')
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {v / cnt[k]:18.1f} per dispatch')
PY
}
: > $out/pmc_summary.txt
pmc tcc FETCH_SIZE
pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pmc tcw WRITE_SIZE
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
cat $out/pmc_summary.txt
