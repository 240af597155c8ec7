#!/bin/bash
# tools/ab_build.sh name "flags" ... : build A/B variants of libaim_hip.so into build_ab/
cd "$(dirname "$0")/.."
mkdir -p build_ab
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Iaim_amd/csrc $f -o build_ab/lib_$n.so aim_amd/csrc/aim_capi.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 'wfa_lane_kernelILi3ELi4ELi1ELi5ELi112' | grep -E 'error|VGPRs:|Scratch' | sed 's/.*remark: *//' | tr '\n' ' '; echo " <- $n" ) &
done
wait
