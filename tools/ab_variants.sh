#!/bin/bash
# tools/ab_variants.sh <case> <frames> <tag> ...: library against each _exp/libimgxf_<tag>.so (tools/build_variant.sh)
case=$1; frames=$2; shift 2
for t in "$@"; do
  echo "B = $t"
  timeout -k 10 200 python tools/ab_lib.py imagetransformations_amd/libimgxf.so _exp/libimgxf_$t.so $case $frames 2>&1 | grep -v amdgpu.ids || exit 1
done
