#!/bin/bash
# Diagnostic helper: build HEAD's engine as tests/cpp/libdbg_prev.so (the "A" of tests/diag_ab.sh).
set -e
rm -rf /tmp/icp_prev && mkdir -p /tmp/icp_prev
git -C "$(dirname "$0")/.." archive HEAD icp_amd/csrc include | tar -x -C /tmp/icp_prev
cd /tmp/icp_prev
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=14 \
    -shared -o "$OLDPWD/tests/cpp/libdbg_prev.so" icp_amd/csrc/*.hip icp_amd/csrc/*.cpp
