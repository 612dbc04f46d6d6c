export AMMC_LIB=$PWD/ammcnet_aaai2021_amd/libammc_hip_stamp.so AMMC_S16_MF=0
python tools/micro/tap_stamps.py 16 128 128 128 128
python tools/micro/tap_stamps.py 16 128 128 64 128
python tools/micro/tap_stamps.py 16 256 256 64 64
python tools/micro/tap_stamps.py 16 64 64 256 256
AMMC_S16_TAP=2 AMMC_S16_MF=1 python tools/micro/tap_stamps.py 16 128 128 128 128
AMMC_S16_TAP=2 AMMC_S16_MF=1 python tools/micro/tap_stamps.py 16 64 64 256 256
AMMC_S16_TAP=2 AMMC_S16_MF=1 python tools/micro/tap_stamps.py 16 32 32 512 512
