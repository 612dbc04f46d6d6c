import sys
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import synthetic as S
from ammcnet_aaai2021_amd.workload import MemoryStress
ms = MemoryStress(S.hashed_normal("stress:e", (512, 8192), 0.9).cuda(), 2)
x = torch.randn(262144, 512, device="cuda") * 0.8
for _ in range(3): ms.run(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ms.run(x)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 100:.1f} us per launch")
