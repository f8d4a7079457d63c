#!/bin/bash
# dev: build/ab/libfrost_<tag>.so = the in-tree objects with ONE source recompiled under extra flags:  mkvariant.sh <tag> <source.hip> <flags...>
set -e
tag=$1; src=$2; shift 2
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c frostnet_amd/csrc/$src -o build/ab/${src%.hip}_$tag.o
objs=""
for o in build/*.o; do b=$(basename $o .o); if [ "$b.hip" = "$src" ]; then objs="$objs build/ab/${b}_$tag.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libfrost_$tag.so $objs
echo built build/ab/libfrost_$tag.so
