AMMC_TAP_KH=1 AMMC_S16_MF=0 python -m pytest tests/test_gpu_conv_tap.py -x -q 2>&1 | tail -5
echo "== base MF0"; AMMC_S16_MF=0 python tools/conv_bench.py --net 16 30 2>&1 | grep -v amdgpu.ids
echo "== KH"; AMMC_TAP_KH=1 AMMC_S16_MF=0 python tools/conv_bench.py --net 16 30 2>&1 | grep -v amdgpu.ids
