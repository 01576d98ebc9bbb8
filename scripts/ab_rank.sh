#!/bin/bash
# Same-box A/B of two builds on the emulated rank of a W-GPU job: scripts/ab_rank.sh <libA.so> <libB.so> [W] [reps]
cd "$(dirname "$0")/.."
W=${3:-8}
reps=${4:-1}
for r in $(seq $reps); do
for lib in "$1" "$2"; do
echo -n "$lib: "
ASL_LIB_PATH=$(readlink -f $lib) python scripts/sim_rank.py $W 2>/dev/null | tail -1
done
done
