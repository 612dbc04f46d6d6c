for s in "32 32 512 512" "32 32 256 512" "64 64 256 256" "64 64 128 256" "128 128 128 128" "128 128 64 128" "256 256 64 64" "256 256 128 64"; do
  python tools/conv_bench.py 16 $s 2>&1 | tail -1
  python tools/conv_bench.py 32 $s 2>&1 | tail -1
done
