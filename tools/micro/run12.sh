for v in "" _pf6 _pf8; do echo "lib$v"; AMMC_LIB=$PWD/ammcnet_aaai2021_amd/libammc_hip$v.so python tools/micro/mt_time.py 2>&1 | tail -1; done
