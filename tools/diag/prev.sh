#!/bin/bash
# Diagnostic helper: build HEAD's engine as tests/cpp/libdbg_prev.so (the "A" of tools/diag/ab.sh).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
rm -rf "$ROOT/build/prev_src" && mkdir -p "$ROOT/build/prev_src"
git -C "$ROOT" archive HEAD icp_amd/csrc include | tar -x -C "$ROOT/build/prev_src"
cd "$ROOT/build/prev_src"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=14 \
    -shared -o "$ROOT/tests/cpp/libdbg_prev.so" icp_amd/csrc/*.hip icp_amd/csrc/*.cpp
