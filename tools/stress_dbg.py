"""Where the time of memory_topk_f16r goes: the kernel's measurement instances (AMMC_F16R_DBG: 1 = no top-K update,
2 = no codebook DMA inside the sweep, 4 = no gather / commit tail, 8 = no feature staging; sums combine), one process
each (the switch is read once per process), 262144 rows.  Results of those instances are WRONG by construction.

    python tools/stress_dbg.py [frames=256]"""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, json, torch
sys.path.insert(0, '.')
from ammcnet_aaai2021_amd import synthetic as S
from ammcnet_aaai2021_amd.workload import MemoryStress
frames = int(sys.argv[1])
dev = "cuda:0"
d, m, k = 512, 8192, 2
n = frames * 1024
embed = S.hashed_normal("stress:e", (d, m), 0.9).to(dev)
g = torch.Generator(device=dev); g.manual_seed(4321)
x = torch.randn(n, d, device=dev, generator=g) * 0.8
ms = MemoryStress(embed, k, rows_in_registers=True)
for _ in range(3): ms.run(x)
torch.cuda.synchronize()
ts = []
for _ in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ms.run(x); e1.record(); torch.cuda.synchronize()
    ts.append(1e3 * e0.elapsed_time(e1))
ts.sort()
print(json.dumps({"us": round(ts[len(ts) // 2], 1), "min": round(ts[0], 1), "tflops": round(ms.flops(n) / ts[len(ts) // 2] / 1e6, 1)}))
'''
frames = sys.argv[1] if len(sys.argv) > 1 else "256"
for dbg in [int(v) for v in os.environ.get("STRESS_DBG_LIST", "0,1,2,4,5,7,15,0").split(",")]:
    env = dict(os.environ, AMMC_F16R_DBG=str(dbg))
    r = subprocess.run([sys.executable, "-c", CHILD, frames], env=env, capture_output=True, text=True)
    print(f"dbg {dbg:2d}:", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
