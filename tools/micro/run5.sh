AMMC_TAP_KH=1 AMMC_S16_MF=0 python -m pytest tests/test_gpu_conv_tap.py -q 2>&1 | tail -30
