#!/bin/bash
# tuning sweep for the 5x5 marching Gaussian (development aid)
for J in 4 0 8; do for RPW in 40 48 54 64 72 90 108; do
  echo "J=$J RPW=$RPW $(WARM=60 ITERS=100 IMGXF_MARCH_J=$J IMGXF_MARCH_RPW=$RPW python tools/bench_ops.py gauss5 64 | head -1)"
done; done
