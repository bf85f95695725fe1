#!/bin/bash
# builds an alternate device library for kernel experiments:  tools/experiments/build_exp.sh <name> [srcdir] [-DFLAG ...]
# -> vecchio_amd/lib/exp/<name>.so (picked up through VK_DEVICE_LIB, see tools/experiments/perf_quick.py)
name=$1; shift
src=${1:-/root/repo}; shift
mkdir -p /root/repo/vecchio_amd/lib/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -Wno-unused-function "$@" \
  -shared -o /root/repo/vecchio_amd/lib/exp/$name.so $src/vecchio_amd/csrc/vk_api.hip $src/vecchio_amd/csrc/vk_linearize.cpp && echo built $name
