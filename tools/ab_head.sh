#!/bin/bash
# A = _exp/libimgxf_head.so (the last commit's build), B = the working tree's library under each given knob set ("-" = none)
case=$1; frames=$2; shift 2
for e in "$@"; do
  echo "B_ENV=$e"
  if [ "$e" = "-" ]; then unset B_ENV; else export B_ENV="$e"; fi
  timeout -k 10 200 python tools/ab_lib.py _exp/libimgxf_head.so imagetransformations_amd/libimgxf.so $case $frames 2>&1 | grep -v amdgpu.ids || exit 1
done
