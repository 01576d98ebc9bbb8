cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r03_ivfflat_np112 --index ivfflat --nprobe 112 > /dev/null 2>&1
python bench.py --workload cascade > gpurun_out/r03_cascade_bench.json 2> gpurun_out/r03_cascade_bench.err
python bench.py --workload cascade --index ivfflat --nprobe 112 > gpurun_out/r03_cascade_ivfflat_np112_bench.json 2>/dev/null
for idx in ivfpq ivfflat; do for W in 2 4 8; do python scripts/sim_rank.py $W 2100000 16384 0 $idx 2>/dev/null | tail -1; done; done > gpurun_out/r03_sim_rank.txt
timeout 600 python scripts/fuzz_paths.py > gpurun_out/r03_fuzz_paths.log 2>&1
tail -3 gpurun_out/r03_fuzz_paths.log
cat gpurun_out/r03_sim_rank.txt
tail -c 400 gpurun_out/r03_cascade_bench.json
