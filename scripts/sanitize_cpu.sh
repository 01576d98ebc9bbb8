#!/bin/bash
# CPU-side sanitizer pass (the reference has none): the oracle and the
# reference's SpectrumMatch.cpp built with AddressSanitizer + UndefinedBehaviorSanitizer
# (`make -C oracle sanitize`), the whole `-m "not gpu"` suite run over them. The C-ABI library itself
# is a HIP build (no GPU sanitizer on this pool); its host-side entry points that the CPU suite calls
# (loading, symbol table, error paths without a device, the .idxmi / FAISS file readers) run under the
# preloaded ASan runtime's interposed allocator too.
#   scripts/sanitize_cpu.sh [log]      (default log: profiles/r06_cpu_sanitizers.log)
set -o pipefail
cd "$(dirname "$0")/.."
log=${1:-profiles/r06_cpu_sanitizers.log}
make -C oracle sanitize > /tmp/san_build.log 2>&1 || { cat /tmp/san_build.log; exit 1; }
asan=$(gcc -print-file-name=libasan.so); ubsan=$(gcc -print-file-name=libubsan.so)
{
  echo "# $(date -u +%FT%TZ)  gcc $(gcc -dumpfullversion)  $(git rev-parse --short HEAD)"
  echo "# LD_PRELOAD=$asan:$ubsan ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1"
  echo "# ASL_ORACLE_SANITIZE=1 python -m pytest tests -q -m 'not gpu' -p no:cacheprovider"
  LD_PRELOAD=$asan:$ubsan ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 \
    UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 ASL_ORACLE_SANITIZE=1 \
    python -m pytest tests -q -m 'not gpu' -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40
  echo "# exit code of pytest: ${PIPESTATUS[0]}"
} | tee "$log"
