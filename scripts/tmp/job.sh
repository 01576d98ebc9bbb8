cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --cpu-seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipelined', d['value'], d['ms_per_step'], d['stages_ms_per_step']); print('fixed_recall', d['fixed_recall']['value'], d['fixed_recall']['ms_per_step'], d['fixed_recall']['scan_ms_per_step'])"
python bench.py --cpu-seconds 0 --no-pipeline --no-fixed-recall --recall-queries 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['value'], d['ms_per_step'], d['stages_ms_per_step'])"
