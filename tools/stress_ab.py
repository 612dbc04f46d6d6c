"""Config 5 A/B in one process: memory_topk_f16r (rows in registers, round 5) against memory_topk_f16 (rows in LDS), the
same rows and codebook, launches interleaved; the two forms' indices / gathers compared with each other and both with the
oracle (bench.stress_parity).

    python tools/stress_ab.py [frames=256] [reps=7]"""
import json
import sys
sys.path.insert(0, '.')
import torch
import bench
from ammcnet_aaai2021_amd import synthetic as S
from ammcnet_aaai2021_amd.workload import MemoryStress

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = "cuda:0"
d, m, k = 512, 8192, 2
n = frames * 1024
embed = S.hashed_normal("stress:e", (d, m), 0.9).to(dev)
g = torch.Generator(device=dev)
g.manual_seed(4321)
x = torch.randn(n, d, device=dev, generator=g) * 0.8
forms = {"f16r": MemoryStress(embed, k, rows_in_registers=True), "f16": MemoryStress(embed, k, rows_in_registers=False)}
res, times = {}, {f: [] for f in forms}
for f, ms in forms.items():
    ms.run(x)
torch.cuda.synchronize()
for _ in range(reps):
    for f, ms in forms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ms.run(x)
        e1.record()
        torch.cuda.synchronize()
        times[f].append(1e3 * e0.elapsed_time(e1))
out = {"rows": n}
for f, ms in forms.items():
    qk, part, q1, idx = ms.run(x)
    torch.cuda.synchronize()
    res[f] = (qk.clone(), part.double().sum().item(), q1.clone(), idx.clone())
    us = sorted(times[f])[len(times[f]) // 2]
    out[f] = {"us": round(us, 1), "tflops": round(ms.flops(n) / us / 1e6, 1), "all_us": [round(t) for t in times[f]],
              "parity": bench.stress_parity(ms, x, qk, idx, d, m, k, rows=16384)}
same = (res["f16r"][3] == res["f16"][3]).all(dim=1)
out["forms_agree_rows"] = float(same.double().mean())
out["gathers_equal_where_agreed"] = bool(torch.equal(res["f16r"][0][same], res["f16"][0][same]))
out["q_one_equal_where_agreed"] = bool(torch.equal(res["f16r"][2][same], res["f16"][2][same]))
out["commit_rel_diff"] = abs(res["f16r"][1] - res["f16"][1]) / res["f16"][1]
print(json.dumps(out))
