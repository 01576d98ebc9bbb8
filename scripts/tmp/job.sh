cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_index.py -x -q 2>&1 | tail -2
for c in 256 512 1024 2048; do
ASL_CS_COVER=$c python bench.py --no-pipeline --cpu-seconds 0 --recall-queries 0 --no-fixed-recall --steps 5 --warmup 1 2>/dev/null | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('cover', $c, 'step', d['ms_per_step'], 'coarse', s['coarse_gemm'])"
done
