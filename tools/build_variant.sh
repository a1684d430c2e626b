#!/bin/bash
# tools/build_variant.sh <tag> <file.hip> [extra hipcc flags...]: _exp/libimgxf_<tag>.so = the library's objects with ONE
# source recompiled under extra flags (ablation / experiment builds for tools/ab_lib.py; development aid)
set -e
tag=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
c=$root/imagetransformations_amd/csrc
mkdir -p $root/_exp/obj
base=$(basename $src .hip)
fileflags=""
[ "$base" = affine ] && fileflags="-fno-slp-vectorize"
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-function -I$root/include -I$c $fileflags "$@" -c $c/$base.hip -o $root/_exp/obj/$base.$tag.o
objs=$(ls $c/_obj/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/_exp/libimgxf_$tag.so $objs $root/_exp/obj/$base.$tag.o
echo built _exp/libimgxf_$tag.so
