#!/bin/bash
# A/B of the shared filter-row reads of the 3x3 weight gradients (AMMC_WGRAD_AF=1, the default: six transposed reads +
# funnel shifts per filter row) against four reads per tap (=0), the network's layer shapes at batch 32, then the step.
for shape in "32 256 256 64 64" "32 256 256 128 64" "32 128 128 128 128" "32 128 128 256 128" "32 64 64 256 256" "32 64 64 512 256" "32 32 32 512 512"; do
  for af in 0 1 0 1; do
    echo -n "af=$af  "
    AMMC_WGRAD_AF=$af python tools/wgrad_bench.py $shape 30 2>&1 | grep -v amdgpu.ids
  done
done
for af in 0 1 0 1; do
  echo "== train step, AMMC_WGRAD_AF=$af"
  AMMC_WGRAD_AF=$af python bench.py --mode train --steps 8 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d.get('ms_per_step'), d.get('train', {}).get('parity', {}).get('ok'))"
done
