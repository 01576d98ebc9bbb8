#!/bin/bash
# One profiling pass for profiles/: bench line, rocprofv3 kernel stats of the same command,
# PMC passes (scripts/pmc.sh). usage: scripts/profile_round.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out/$tag
python3 bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o x -- python3 bench.py --cpu-seconds 0 > gpurun_out/$tag/bench_under_rocprof.json 2> /tmp/prof_$tag.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/$tag/kernel_stats.csv
bash scripts/pmc.sh $tag/pmc > /dev/null 2>&1
cp gpurun_out/$tag/pmc/summary.txt gpurun_out/$tag/pmc_summary.txt
tail -c 1500 gpurun_out/$tag/bench.json
