#!/bin/bash
# A/B helper: build the library of a git revision (default HEAD) into build_ab/lib_<name>.so next to the working tree's
# (load it with AIM_LIB=build_ab/lib_<name>.so).  tools/ab_base.sh [rev] [name]
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}; name=${2:-base}
src=build_ab/src_$name; obj=build/obj_ab_$name
rm -rf "$src"; mkdir -p "$src" "$obj"
git archive "$rev" aim_amd/csrc include | tar -x -C "$src"
ls "$src"/aim_amd/csrc/*.hip | xargs -P 8 -I{} sh -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I'"$src"'/include -I'"$src"'/aim_amd/csrc -c {} -o '"$obj"'/$(basename {} .hip).o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/lib_$name.so "$obj"/*.o
echo built build_ab/lib_$name.so from $rev
