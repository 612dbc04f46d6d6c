run() { echo "== $1"; env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', d['value'], d['ms_per_step'], d['parity_max_rel']); print('   ', ' | '.join(f\"{k.replace('conv_','')} {v['launches_per_step']}x{v['avg_us']}\" for k,v in d['kernels'].items() if v['share']>0.02))"; }
for rep in 1 2; do
run "AMMC_TAP_KH=0"
run "AMMC_TAP_KH=0 AMMC_S16_MF=1"
run "AMMC_TAP_KH=0 AMMC_S16_MF=1 AMMC_S16_TAP=2"
run "AMMC_TAP_KH=1"
done
