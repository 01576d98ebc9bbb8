cd $GRAFT_REPO_ROOT
timeout 1500 python scripts/fuzz_paths.py 1400 77 > gpurun_out/r03_fuzz_paths_long.log 2>&1; tail -2 gpurun_out/r03_fuzz_paths_long.log
timeout 500 python scripts/fuzz_shards.py 400 5 > gpurun_out/r03_fuzz_shards2.log 2>&1; tail -1 gpurun_out/r03_fuzz_shards2.log
