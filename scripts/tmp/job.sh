cd $GRAFT_REPO_ROOT
NPROBE=112 REPS=2 bash scripts/ab_flat.sh scripts/tmp/flatA.so scripts/tmp/flatD.so scripts/tmp/flatF.so > gpurun_out/r3_flat_ab3.txt 2>&1
cat gpurun_out/r3_flat_ab3.txt
