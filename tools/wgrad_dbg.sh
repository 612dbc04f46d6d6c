#!/bin/bash
# What bounds the 3x3 weight-gradient kernel: AMMC_WGRAD_DBG ablations (wrong results, timing only) on three layer shapes.
#   0 = the kernel, 1 = no DMA in the patch loop, 2 = no contraction (DMA + barriers only), 4 = no epilogue, 6 = DMA only, no stores
for shape in "32 256 256 64 64" "32 128 128 128 128" "32 64 64 256 256"; do
  for dbg in 0 1 2 4 5; do
    echo -n "dbg=$dbg  "
    AMMC_WGRAD_DBG=$dbg python tools/wgrad_bench.py $shape 30 2>&1 | grep -v amdgpu.ids
  done
done
