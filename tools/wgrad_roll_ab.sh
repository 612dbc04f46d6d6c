#!/bin/bash
# A/B of the rolling-halo form of the 3x3 weight gradients (AMMC_WGRAD_ROLL=1, the default) against the per-patch halo
# (=0) on the network's layer shapes at the timed training batch (32 clips), then the whole training step.
for shape in "32 256 256 64 64" "32 256 256 128 64" "32 128 128 128 128" "32 128 128 256 128" "32 64 64 256 256" "32 64 64 512 256" "32 32 32 512 512"; do
  for roll in 0 1 0 1; do
    echo -n "roll=$roll  "
    AMMC_WGRAD_ROLL=$roll python tools/wgrad_bench.py $shape 30
  done
done
for roll in 0 1 0 1; do
  echo "== train step, AMMC_WGRAD_ROLL=$roll"
  AMMC_WGRAD_ROLL=$roll python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d.get('ms_per_step'), (d.get('parity') or {}).get('ok'), ((d.get('parity') or {}).get('vs_fp64') or {}).get('ok'))"
done
