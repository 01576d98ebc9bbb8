#!/bin/bash
# PMC passes over the bench (only this repo's hot kernels); writes small CSV summaries.
# usage: scripts/pmc.sh <tag> [bench args...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "pq_scan|flat_inv|rescore_|gemm_nt|row_topk|row_select|coarse_sparse" \
     --output-format csv -d /tmp/pmc_$name -o x -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --recall-queries 0 --no-cascade --no-reference-geometry --no-recall-hard --beyond-llc-chunks 0 "${BENCH_ARGS[@]}" > /tmp/pmc_$name.log 2>&1
  python3 - "$name" <<'PY' >> gpurun_out/$TAG/summary.txt
import csv, sys, collections, glob
name = sys.argv[1]
f = glob.glob(f'/tmp/pmc_{name}/**/x_counter_collection.csv', recursive=True)
if not f:
    print(name, 'no counter file'); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    # launches of one kernel at different sizes (the bench also runs half-size batches and small samples)
    # are different rows: a per-dispatch average must not mix them
    k = r['Kernel_Name'].split('(')[0][:60] + ' grid=' + str(r.get('Grid_Size', '?'))
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in acc:
    print(f'[{name}] {k} dispatches={cnt[k]}')
    for c, v in sorted(acc[k].items()):
        print(f'    {c:28s} {v / cnt[k]:16.1f} per dispatch')
PY
}
export TAG=$tag
BENCH_ARGS=("$@")
: > gpurun_out/$tag/summary.txt
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_SCA
run tcc FETCH_SIZE
# (TA_* counters hang the profiler on this pool: not collected)
if [ -n "$PMC_EXTRA" ]; then
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum
run sq3 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VALU
fi
run tcw WRITE_SIZE
cat gpurun_out/$tag/summary.txt
