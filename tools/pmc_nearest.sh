#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the NEAREST rotation kernels (development aid; one pass per counter and kernel variant)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in wq old; do
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ $v = old ]; then export IMGXF_AFFINE_NO_WQ=1; else unset IMGXF_AFFINE_NO_WQ; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_nearest/$v.$c -- python3 $R/tools/nearest_once.py "$@" > $R/gpurun_out/pmc_nearest.$v.$c.log 2>&1 || exit 1
  done
  echo "== $v"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_nearest/$v.FETCH_SIZE nearest; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_nearest/$v.WRITE_SIZE nearest
done
