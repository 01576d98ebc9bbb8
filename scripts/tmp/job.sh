cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
for lib in scripts/tmp/base.so ann_solo_amd/libannsolo_mi.so; do
ASL_LIB_PATH=$PWD/$lib python bench.py --no-pipeline --cpu-seconds 0 --recall-queries 0 --no-fixed-recall --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('$lib pq', 'step', d['ms_per_step'], 'scan', s['scan'])"
ASL_LIB_PATH=$PWD/$lib python bench.py --index ivfflat --nprobe 112 --no-pipeline --cpu-seconds 0 --recall-queries 0 --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('$lib flat', 'step', d['ms_per_step'], 'scan', s['scan'])"
done; done
bash scripts/ab_rank.sh scripts/tmp/base.so ann_solo_amd/libannsolo_mi.so 8 1
