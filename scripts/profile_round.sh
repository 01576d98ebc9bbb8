#!/bin/bash
# One profiling pass for profiles/: bench line, rocprofv3 kernel stats of the same command,
# PMC passes (scripts/pmc.sh). usage: scripts/profile_round.sh <tag> [bench.py arguments...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
python3 bench.py "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o x -- python3 bench.py --cpu-seconds 0 --no-reference-geometry "$@" > gpurun_out/$tag/bench_under_rocprof.json 2> /tmp/prof_$tag.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/$tag/kernel_stats.csv
# the default run launches the scan at several sizes (recall sample, cascade tail): the per-dispatch
# trace grouped by grid size gives the FULL-SIZE launches' average that roofline.avg_launch_ms must match
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 - "$t" > gpurun_out/$tag/scan_launches.txt <<'PY'
import csv, sys, collections
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'pq_scan_v3_kernel' in n or 'flat_inv_scan_kernel' in n:
        wg = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
        g[(n.split('(')[0].replace('void asl::', ''), wg)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
print('scan-kernel dispatches of the run under rocprofv3 --kernel-trace, by kernel and workgroups (= queries) per launch')
for (n, wg), v in sorted(g.items()):
    print(f'{n:45s} queries {wg:6d}  launches {len(v):3d}  avg {sum(v) / len(v):7.3f} ms  min {min(v):7.3f}  max {max(v):7.3f}')
PY
bash scripts/pmc.sh $tag/pmc "$@" > /dev/null 2>&1
cp gpurun_out/$tag/pmc/summary.txt gpurun_out/$tag/pmc_summary.txt
# HBM-side bytes of the dominant scan kernel per launch (FETCH_SIZE is in KiB; x2 on gfx950 for a
# wide coalesced stream, MI355X_MICROARCH.md) -> the constant bench.py reports as roofline.traffic
python3 - gpurun_out/$tag/pmc_summary.txt gpurun_out/$tag/pmc_traffic.json "$tag" "$@" <<'PY'
import json, re, sys
src, dst, tag = sys.argv[1:4]
cur, vals = None, {}
for line in open(src):
    m = re.match(r'\[(\w+)\] (.*) dispatches=(\d+)', line)
    if m:
        cur = m.group(2).strip()
        continue
    m = re.match(r'\s+(\w+)\s+([0-9.]+) per dispatch', line)
    if m and cur and ('pq_scan_v3' in cur or 'flat_inv_scan' in cur):
        vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
out = {}
def grid(k):
    m = re.search(r'grid=(\d+)', k)
    return int(m.group(1)) if m else 0
# the full-size launches = the largest grid of the scan kernel (smaller ones: half-size batches, samples)
for k, v in sorted(vals.items(), key=lambda kv: -grid(kv[0])):
    if 'FETCH_SIZE' in v:
        sys.path.insert(0, '.')
        import bench
        out = {'kernel': re.sub(r' grid=\d+', '', k).replace('void asl::', ''), 'queries_per_launch': grid(k) // 512,
               'kernel_source_sha1': bench.kernel_source_sha1('pq' if 'pq_scan' in k else 'flat'),
               'workload': 'bench.py ' + (' '.join(sys.argv[4:]) or 'defaults (2.1M library, nlist 4096, nprobe 128, k 1024, 32768 queries)'),
               'FETCH_SIZE_KiB_per_dispatch': v['FETCH_SIZE'],
               'WRITE_SIZE_KiB_per_dispatch': v.get('WRITE_SIZE'),
               'correction': 'x2 on FETCH_SIZE (gfx950, wide coalesced 16-B/lane stream; MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected',
               'scan_hbm_bytes_per_launch': int(v['FETCH_SIZE'] * 1024 * 2 + (v.get('WRITE_SIZE') or 0) * 1024),
               'source': f'profiles/{tag}_pmc_summary.txt (scripts/pmc.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'}
        break
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps(out))
PY
tail -c 1500 gpurun_out/$tag/bench.json
