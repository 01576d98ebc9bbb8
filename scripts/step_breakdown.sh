#!/bin/bash
# Kernel-by-kernel picture of ONE pipelined bench step (stream B: scan + rescoring; stream A: front):
# rocprofv3 --kernel-trace over a short bench, the launches of the timed loop grouped by kernel
# and grid size.   scripts/step_breakdown.sh <tag> [bench args]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/sb_$tag -o x -- python3 bench.py --steps 8 --warmup 2 --cpu-seconds 0 --recall-queries 0 --no-cascade --no-reference-geometry --no-recall-hard --no-fixed-recall --beyond-llc-chunks 0 "$@" > gpurun_out/$tag/bench.json 2> /tmp/sb_$tag.err
t=$(find /tmp/sb_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$t" > gpurun_out/$tag/step_breakdown.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
scan = [r for r in rows if 'pq_scan_v3_kernel' in r['Kernel_Name'] or 'flat_inv_scan_kernel' in r['Kernel_Name']]
full = max(int(r['Grid_Size_X']) for r in scan)
scan = [r for r in scan if int(r['Grid_Size_X']) == full]
# the timed loop = the last 8 full-size scans of the FIRST contiguous run of full-size scans (warm-up 2 + 8 steps)
first = scan[:10]
t0, t1 = int(first[2]['Start_Timestamp']), int(first[9]['End_Timestamp'])
# extend to the end of the rescoring that follows the last scan: next full-size scan start or +5 ms
nxt = [int(r['Start_Timestamp']) for r in scan[10:11]]
t1 = nxt[0] if nxt else t1 + 5_000_000
sel = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
g = collections.defaultdict(list)
for r in sel:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('asl::', '')[:52]
    g[(name, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), r.get('Stream_Id', r.get('Queue_Id', '?')))].append(
        (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
print(f'window {(t1 - t0) / 1e6:.2f} ms = 8 steps -> {(t1 - t0) / 8e6:.3f} ms per step; kernels by (name, workgroups, queue): launches, ms per step, avg ms')
tot = 0
for (name, wg, q), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f'{name:52s} wg {wg:7d} q {q:>4s}  n {len(v):3d}  {sum(v) / 8:7.3f} ms/step  avg {sum(v) / len(v):7.3f}')
    tot += sum(v)
print(f'sum of kernel durations per step: {tot / 8:.3f} ms')
PY
cat gpurun_out/$tag/step_breakdown.txt
