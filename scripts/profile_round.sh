#!/bin/bash
# One profiling pass for profiles/: bench line, rocprofv3 kernel stats of the same command,
# PMC passes (scripts/pmc.sh). usage: scripts/profile_round.sh <tag> [bench.py arguments...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
python3 bench.py "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o x -- python3 bench.py --cpu-seconds 0 "$@" > gpurun_out/$tag/bench_under_rocprof.json 2> /tmp/prof_$tag.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/$tag/kernel_stats.csv
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
for k, v in vals.items():
    if 'FETCH_SIZE' in v:
        out = {'kernel': k.replace('void asl::', ''),
               'workload': 'bench.py ' + (' '.join(sys.argv[4:]) or 'defaults (2.1M library, nlist 4096, nprobe 128, k 1024, 16384 queries)'),
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
