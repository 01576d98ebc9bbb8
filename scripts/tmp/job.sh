cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for r in 1 2; do
python bench.py --no-pipeline --cpu-seconds 0 --recall-queries 0 --no-fixed-recall --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('serial step', d['ms_per_step'], s)"
done
python bench.py --cpu-seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipelined', d['value'], d['ms_per_step'], d['stages_ms_per_step']); print('fixed_recall', d['fixed_recall']['value'], d['fixed_recall']['ms_per_step'], d['fixed_recall']['scan_ms_per_step'])"
