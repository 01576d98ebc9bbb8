cd $GRAFT_REPO_ROOT
python bench.py --index ivfflat --no-pipeline --cpu-seconds 0 --recall-queries 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('flat np128 serial', d['value'], d['ms_per_step'], d['stages_ms_per_step'])"
python bench.py --index ivfflat --cpu-seconds 0 --recall-queries 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('flat np128 pipelined', d['value'], d['ms_per_step'], d['stages_ms_per_step'])"
