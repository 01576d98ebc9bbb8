#!/bin/bash
# Build a VARIANT of libannsolo_mi.so for same-box A/B runs (ASL_LIB_PATH, scripts/ab_*.sh):
#
#   scripts/build_variant.sh <out.so> [NAME=VALUE ...] [sed:<file>:<expression> ...] [-- extra hipcc flags]
#
# `sed:flat_scan.hip:s/a/b/` applies a sed expression to one file of the COPY (timing-only
# experiments that are not a constant).
#
# The product sources carry their tuning values as plain `constexpr` constants; a variant is
# made from a COPY of csrc/ (scripts/tmp/variant_<hash>/csrc, git-ignored) in which every
# `constexpr ... NAME = <old>` named on the command line is rewritten to the given value,
# compiled into that copy's own build/ directory. The product's sources, objects and library
# are never touched, so an incremental product build cannot pick a variant's objects up.
# The overrides and flags are appended to <copy>/build/build.log.
set -e
cd "$(dirname "$0")/.."
out=$(readlink -f "$1"); shift
defs=(); extra=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi
  defs+=("$1"); shift
done
tag=$(echo "${defs[*]} ${extra[*]}" | md5sum | cut -c1-10)
root=scripts/tmp/variant_$tag
rm -rf "$root"; mkdir -p "$root/ann_solo_amd" "$root/include"
cp -r ann_solo_amd/csrc "$root/ann_solo_amd/csrc"; rm -rf "$root/ann_solo_amd/csrc/build"
cp include/annsolo_mi.h "$root/include/"
for d in "${defs[@]}"; do
  if [[ "$d" == sed:* ]]; then
    rest=${d#sed:}; file=${rest%%:*}; expr=${rest#*:}
    before=$(md5sum < "$root/ann_solo_amd/csrc/$file")
    sed -i -E "$expr" "$root/ann_solo_amd/csrc/$file"
    [ "$before" != "$(md5sum < "$root/ann_solo_amd/csrc/$file")" ] || { echo "build_variant: '$expr' changed nothing in $file" >&2; exit 1; }
    continue
  fi
  name=${d%%=*}; val=${d#*=}
  hits=$(grep -lE "constexpr .*\\b$name = [^,;]+[,;]" "$root"/ann_solo_amd/csrc/*.h* || true)
  [ -n "$hits" ] || { echo "build_variant: no 'constexpr ... $name = ...' in csrc/" >&2; exit 1; }
  sed -i -E "/constexpr/ s/(\\b$name = )[^,;]+([,;])/\\1$val\\2/" $hits
done
make -C "$root/ann_solo_amd/csrc" -j8 EXTRA="${extra[*]}" OUT="$out" > "$root/make.log" 2>&1 || { tail -20 "$root/make.log" >&2; exit 1; }
echo "$(date +%FT%T) variant ${defs[*]} EXTRA='${extra[*]}' -> $out" >> "$root/ann_solo_amd/csrc/build/build.log"
echo "built $out with ${defs[*]} ${extra[*]} (sources: $root)"
