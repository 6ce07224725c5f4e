#!/bin/bash
# A second build of libbtrapz_hip.so with extra compiler flags, for A/B runs of kernel variants on one GPU box:
#   tools/build_variant.sh B -DLEAN_E_CACHE=0     ->  scratch/variants/B/libbtrapz_hip.so
#   BTRAPZ_HIP_LIB=scratch/variants/B/libbtrapz_hip.so python tools/lean_bench.py
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/scratch/variants/$name
mkdir -p "$out"
make -s -j4 -C "$root/spectral_amd/csrc" OUT="$out" OBJDIR="$out/obj" HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $*" "$out/libbtrapz_hip.so"
ls -la "$out/libbtrapz_hip.so"
