python -m pytest tests/test_gpu_conv_tap.py tests/test_gpu_s16.py -x -q 2>&1 | tail -3
python tools/conv_bench.py --net 16 30 2>&1 | grep -v amdgpu.ids
export AMMC_LIB=$PWD/ammcnet_aaai2021_amd/libammc_hip_stamp.so
AMMC_S16_MF=0 python tools/micro/tap_stamps.py 16 128 128 128 128 2>&1 | grep -v amdgpu.ids
AMMC_S16_MF=0 python tools/micro/tap_stamps.py 16 256 256 64 64 2>&1 | grep -v amdgpu.ids
