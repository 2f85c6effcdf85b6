#!/bin/bash
# developer aid: build a variant library that differs from the production one in ONE object (default dg_corr2.o)
#   [SRC=dg_corr] scripts/build_variant.sh <tag> <extra flags...>   ->  depthg_amd/lib/libdepthg_<tag>.so   (select with DEPTHG_LIB)
set -e
tag=$1; shift
src=${SRC:-dg_corr2}
cd "$(dirname "$0")/../depthg_amd/csrc"
mkdir -p ../lib/obj_$tag
[ "$src" = dg_corr2 ] && set -- -mllvm -disable-machine-licm "$@"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $src.hip -o ../lib/obj_$tag/$src.o
objs=$(ls ../lib/obj/*.o | grep -v "/$src.o")
hipcc -shared -fPIC --offload-arch=gfx950 $objs ../lib/obj_$tag/$src.o -o ../lib/libdepthg_$tag.so
echo built ../lib/libdepthg_$tag.so
