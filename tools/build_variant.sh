#!/bin/bash
# Build an experimental copy of libpvnet_vote.so into build/variants/<name>.so from a patched scratch copy of csrc/
# (the product sources stay untouched).  usage: tools/build_variant.sh <name> [-e sed-expr]... [--py edit.py] [-Dmacro]... [-X "extra flags"]...
#   tools/build_variant.sh base
#   tools/build_variant.sh nohot -e 's/ht < nht; ++ht) {\s*$/ht < 0; ++ht) {/'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
scratch=$(mktemp -d /tmp/pvv_variant.XXXXXX)
cp "$ROOT"/clean-pvnet_amd/csrc/*.hpp "$ROOT"/clean-pvnet_amd/csrc/pvnet_vote.hip "$scratch"/
sedargs=(); defs=()
while [ $# -gt 0 ]; do
  case "$1" in
    -e) sedargs+=(-e "$2"); shift 2;;
    -D*) defs+=("$1"); shift;;
    -X) defs+=($2); shift 2;;                    # extra compiler flags, e.g. -X "-mllvm -amdgpu-sched-strategy=max-ilp"
    --py) python "$2" "$scratch"; shift 2;;      # a python script that edits the scratch copy in place
    *) echo "unknown arg $1"; exit 1;;
  esac
done
if [ ${#sedargs[@]} -gt 0 ]; then sed -i -E "${sedargs[@]}" "$scratch"/*.hpp "$scratch"/*.hip; fi
mkdir -p "$ROOT/build/variants"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -fPIC -shared \
  -fvisibility=hidden "${defs[@]}" -I"$ROOT/include" -I"$scratch" -o "$ROOT/build/variants/$name.so" "$scratch/pvnet_vote.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -S --cuda-device-only \
  "${defs[@]}" -I"$ROOT/include" -I"$scratch" -o "$ROOT/build/variants/$name.s" "$scratch/pvnet_vote.hip" 2>/dev/null
grep -E "k_count_bf16.*\.(num_vgpr|private_seg_size)," "$ROOT/build/variants/$name.s" | sed 's/.*Consts\w*\././'
diff -ru "$ROOT/clean-pvnet_amd/csrc" "$scratch" --exclude="*.cpp" | grep '^[+-]' | grep -v '^+++\|^---' || true
rm -rf "$scratch"
