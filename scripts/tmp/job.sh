cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_index.py tests/test_gpu_search.py tests/test_gpu_faiss_file.py -x -q 2>&1 | tail -3
python bench.py --cpu-seconds 0 --recall-queries 0 --no-fixed-recall --steps 5 2>&1 | grep -E "built in|^\{" | cut -c1-200
python bench.py --index ivfflat --nprobe 112 --cpu-seconds 0 --recall-queries 0 --steps 5 2>&1 | grep -E "built in" | cut -c1-200
