export AMMC_LIB=$PWD/ammcnet_aaai2021_amd/libammc_hip_stamp.so
STAMP_RAW=1 AMMC_S16_MF=0 python tools/micro/tap_stamps.py 16 128 128 128 128 2>&1 | grep -E "kernel|tile "
