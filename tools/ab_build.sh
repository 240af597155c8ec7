#!/bin/bash
# tools/ab_build.sh name "flags" ... : build A/B variants of libaim_hip.so into build_ab/lib_<name>.so (objects in build/obj_<name>/)
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  python -m aim_amd.build --variant "$n" --flags "$f" || exit 1
done
