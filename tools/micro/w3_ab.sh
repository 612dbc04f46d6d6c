#!/bin/bash
# A/B of two builds of the library on the 3x3 weight-gradient layers of the 256x256 network at batch 32
for rep in 1 2; do
for lib in "" ammcnet_aaai2021_amd/libammc_hip_w3old.so; do
  echo "== lib=${lib:-default} rep=$rep"
  for shape in "32 128 128 128 128" "32 128 128 64 128" "32 64 64 256 256" "32 256 256 64 64" "32 32 32 512 512" "32 128 128 256 128" "32 256 256 128 64"; do
    AMMC_LIB=$lib python tools/wgrad_bench.py $shape 20 2>&1 | tail -1
  done
done
done
