"""frames/s of the eval forward vs batch size (wall clock, inputs resident): python tools/batch_sweep.py [precision]"""
import sys, time
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
prec = sys.argv[1] if len(sys.argv) > 1 else "s16"
net = A.get_twostream((12, 6), (3, 2), 64, 2000, 2)
net.load_state_dict(S.make_twostream_state(n_embed=2000))
net = net.cuda().eval(); net.precision = prec
for B in (1, 2, 4, 8, 16, 32):
    rgb, op, _, _ = S.make_clips(B, 256, 256, tag="bs")
    rgb, op = rgb.cuda(), op.cuda()
    for _ in range(3): net(rgb, op)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = max(5, 64 // B)
    for _ in range(n): net(rgb, op)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{prec} B={B:3d}  {dt*1e3:8.3f} ms/step  {B/dt:8.1f} frames/s")
