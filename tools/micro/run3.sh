export AMMC_LIB=$PWD/ammcnet_aaai2021_amd/libammc_hip_stamp.so AMMC_S16_MF=0
for kh in 0 1; do echo "== KH=$kh"; AMMC_TAP_KH=$kh STAMP_RAW=1 python tools/micro/tap_stamps.py 16 128 128 128 128 2>&1 | grep -v amdgpu.ids; done
