export AMMC_S16_MF=0
echo "== base"; python tools/conv_bench.py --net 16 30 2>&1 | grep -v amdgpu.ids
for dl in 0 400 800 1200 1600; do echo "== PERS delay=$dl"; AMMC_TAP_PERS=1 AMMC_TAP_DELAY=$dl python tools/conv_bench.py --net 16 30 2>&1 | grep -v amdgpu.ids; done
