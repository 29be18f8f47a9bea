#!/bin/bash
# Builds the HIP library of ANOTHER git revision beside the current one, for same-box A/B timing (tools/ab_attn.py):
#   tools/ab_build.sh <rev>   ->  llama-x_amd/llx/libllx_hip_prev.so   (git-ignored like every .so; travels to the GPU box)
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/llx_ab.XXXXXX)
git -C "$root" archive "$rev" llama-x_amd/csrc | tar -x -C "$tmp"
make -C "$tmp/llama-x_amd/csrc" -j8 OUT="$tmp/libllx_hip_prev.so" > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
cp "$tmp/libllx_hip_prev.so" "$root/llama-x_amd/llx/libllx_hip_prev.so"
rm -rf "$tmp"
echo "built llama-x_amd/llx/libllx_hip_prev.so from $rev"
