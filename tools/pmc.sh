#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> "<counters pass1>" ["<counters pass2>" ...] -- <python script args...>
# Collects rocprofv3 PMC counters (one pass per counter group) for a python command.
out=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done; shift
cd /tmp; export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out/pass$i -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out.pass$i.log 2>&1 || echo "pass $i failed"
done
