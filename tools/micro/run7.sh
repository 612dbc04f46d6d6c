for kh in 0 1 2; do echo "== KH=$kh"; AMMC_TAP_KH=$kh python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_max_rel']); [print('   ',k,v['launches_per_step'],v['avg_us'],v['share']) for k,v in d['kernels'].items() if v['share']>0.01]"; done
for kh in 0 1; do echo "== KH=$kh again"; AMMC_TAP_KH=$kh python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_max_rel'])"; done
