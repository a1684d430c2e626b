#!/bin/bash
# A/B of one library against itself under different knobs: tools/ab_env.sh <case> <frames> "<A_ENV>" "<B_ENV>" ...
case=$1; frames=$2; shift 2
base="$1"; shift
for e in "$@"; do
  echo "A_ENV=$base B_ENV=$e"
  A_ENV="$base" B_ENV="$e" timeout -k 10 200 python tools/ab_lib.py imagetransformations_amd/libimgxf.so imagetransformations_amd/libimgxf.so $case $frames 2>&1 | grep -v amdgpu.ids || exit 1
done
