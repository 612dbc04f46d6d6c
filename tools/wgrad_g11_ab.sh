#!/bin/bash
# EXPERIMENT (round 5): the gradient operand of the 3x3 weight gradients with its 11-bit hi half only
# (AMMC_WGRAD_G11=1: two MFMAs per product block instead of three).  Time per layer and per step, and what it does to the
# batch-32 gradients against the fp64 truth (tools/grad_truth.py; the test's gates).
for shape in "32 256 256 64 64" "32 128 128 128 128" "32 64 64 256 256" "32 32 32 512 512"; do
  for g in 0 1 0 1; do
    echo -n "g11=$g  "
    AMMC_WGRAD_G11=$g python tools/wgrad_bench.py $shape 30 2>&1 | grep -v amdgpu.ids
  done
done
for g in 0 1 0 1; do
  echo "== train step, AMMC_WGRAD_G11=$g"
  AMMC_WGRAD_G11=$g python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d.get('ms_per_step'), (d.get('parity') or {}).get('ok'), ((d.get('parity') or {}).get('vs_fp64') or {}).get('ok'))"
done
for g in 0 1; do
  echo "== gradients against the fp64 truth, AMMC_WGRAD_G11=$g"
  AMMC_WGRAD_G11=$g python tools/grad_truth.py --precisions s16 2>&1 | grep -v amdgpu.ids | head -30
  AMMC_WGRAD_G11=$g python -m pytest tests/test_gpu_train.py -q -x -k test_batch32_gradients_against_the_fp64_truth 2>&1 | tail -4
done
