#!/bin/bash
# Build attention.hip (+ the error buffer) alone into tools/_ab/libattn_<name>.so for one-process A/B runs with
# tools/attn_bench.py (ATTN_AB_LIBS=path[,path...]):   tools/build_attn_variant.sh <name> [-DFLAG ...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
mkdir -p "$ROOT/tools/_ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -shared \
  "$ROOT/bagel_amd/csrc/attention.hip" "$ROOT/bagel_amd/csrc/capi.hip" -o "$ROOT/tools/_ab/libattn_$name.so"
echo "$ROOT/tools/_ab/libattn_$name.so"
