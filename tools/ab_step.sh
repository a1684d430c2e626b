#!/bin/bash
# A = library with IMGXF_MFMA_V2=1 (sepconv_mfma2_rgb_kernel), B = each given library (or the library itself) on the k = 31 Gaussian
frames=$1; shift
for t in "$@"; do
  lib=_exp/libimgxf_$t.so; [ "$t" = lib ] && lib=imagetransformations_amd/libimgxf.so
  echo "B = $t"
  A_ENV=IMGXF_MFMA_V2=1 timeout -k 10 200 python tools/ab_lib.py imagetransformations_amd/libimgxf.so $lib gaussian31 $frames 2>&1 | grep -v amdgpu.ids || exit 1
done
