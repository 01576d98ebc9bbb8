#!/bin/bash
# Build a variant of libannsolo_mi.so for same-box A/B runs: scripts/build_variant.sh <out.so> "<-D flags>"
# (objects are shared with the product build: the product is rebuilt with -B afterwards)
cd "$(dirname "$0")/.."
out=$(readlink -f "$1"); shift
make -C ann_solo_amd/csrc -B -j8 EXTRA="$*" OUT="$out" > /dev/null && echo "built $out with $*"
