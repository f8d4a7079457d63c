#!/bin/bash
# dev: build/var/libfrost_<tag>.so = the WHOLE library rebuilt under extra flags (header-level -D overrides):  mkfull.sh <tag> <flags...>
set -e
tag=$1; shift
mkdir -p build/var build/var_$tag
for s in frostnet_amd/csrc/*.hip; do
  b=$(basename $s .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -w "$@" -c $s -o build/var_$tag/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/libfrost_$tag.so build/var_$tag/*.o
rm -rf build/var_$tag
echo built build/var/libfrost_$tag.so
