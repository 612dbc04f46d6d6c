for v in 0 2 3 4; do
  echo "== dbg $v"
  export AMMC_S16_DBG=$v
  python tools/conv_bench.py 16 256 256 64 64 2>&1 | tail -1
  python tools/conv_bench.py 16 256 256 128 64 2>&1 | tail -1
  python tools/conv_bench.py 16 128 128 128 128 2>&1 | tail -1
  python tools/conv_bench.py 16 128 128 256 128 2>&1 | tail -1
done
