for kh in 2 1; do echo "== KH=$kh"; AMMC_TAP_KH=$kh python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | tail -4 | cut -c1-900; done
