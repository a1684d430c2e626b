#!/bin/bash
# Sweep the rows-per-chunk of the marching Gaussian (development aid).
for RPW in 48 64 94 128 188; do
  echo "RPW=$RPW $(WARM=60 ITERS=100 IMGXF_MARCH_RPW=$RPW python tools/bench_ops.py gauss5 64 | head -1)"
done
