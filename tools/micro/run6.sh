export AMMC_S16_MF=0
echo "== base"; python tools/conv_bench.py --net 16 30 2>&1 | grep -v amdgpu.ids
export AMMC_TAP_KH=1
for v in "" _khb; do echo "== lib$v"; AMMC_LIB=$PWD/ammcnet_aaai2021_amd/libammc_hip$v.so python -m pytest tests/test_gpu_conv_tap.py -q -x 2>&1 | tail -1;  AMMC_LIB=$PWD/ammcnet_aaai2021_amd/libammc_hip$v.so python tools/conv_bench.py --net 16 30 2>&1 | grep -v amdgpu.ids; done
AMMC_LIB=$PWD/ammcnet_aaai2021_amd/libammc_hip_stamp.so STAMP_RAW=1 python tools/micro/tap_stamps.py 16 128 128 128 128 2>&1 | grep -v amdgpu.ids
