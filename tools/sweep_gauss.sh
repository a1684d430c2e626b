#!/bin/bash
# tuning sweep for the 5x5 marching Gaussian (development aid)
for J in 3 4 6 8 12; do for RPW in 32 64 135 270; do
  echo "J=$J RPW=$RPW $(IMGXF_MARCH_J=$J IMGXF_MARCH_RPW=$RPW python tools/bench_ops.py gauss5 64 | head -1)"
done; done
