"""where a workgroup of the fused memory-block kernel spends its cycles (AMMC_MB_STAMPS=1): s_memtime at the phase boundaries"""
import os, sys, ctypes as C
os.environ["AMMC_MB_STAMPS"] = "1"
sys.path.insert(0, ".")
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S, _lib
lib = _lib.load()
net = A.get_twostream((12, 6), (3, 2), 64, 2000, 2)
net.load_state_dict(S.make_twostream_state(n_embed=2000))
net = net.to("cuda:0").eval()
rgb_x, op_x, _, _ = (t.to("cuda:0") for t in S.make_clips(16, 256, 256, tag="bench"))
net._engine = None
with torch.no_grad():
    for _ in range(3):
        net(rgb_x, op_x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
fn = lib.ammc_dbg_memory_block_stamps
print("rc", fn(buf))
names = ["start", "A enc done", "norms", "B sweep done", "cand barrier", "merge done", "C gather+reduce done", "D mfma done", "D epilogue done", "end"]
for wg in range(2):
    t = list(buf)[16 * wg:16 * wg + 10]
    print("wg", wg, [(names[i + 1], t[i + 1] - t[i]) for i in range(9)], "total", t[9] - t[0])
