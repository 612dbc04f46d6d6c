for tap in 1 2 4; do for mf in 0 1; do echo "== TAP=$tap MF=$mf"; AMMC_S16_TAP=$tap AMMC_S16_MF=$mf python tools/conv_bench.py --net 16 30 2>&1 | grep -v amdgpu.ids; done; done
