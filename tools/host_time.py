import sys, time
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
net = A.get_twostream((12, 6), (3, 2), 64, 2000, 2)
net.load_state_dict(S.make_twostream_state(n_embed=2000))
net = net.cuda().eval()
rgb, op, _, _ = S.make_clips(16, 256, 256, tag="lt")
rgb, op = rgb.cuda(), op.cuda()
for guard in (True, False):
    net.s16_guard = guard
    with torch.no_grad():
        for _ in range(5): net(rgb, op)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 20
        host = 0.0
        for _ in range(n):
            h0 = time.perf_counter()
            net(rgb, op)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
    print(f"guard={guard}: wall {tot/n*1e3:.3f} ms/step, host time inside forward() {host/n*1e3:.3f} ms/step")
# pure enqueue cost: guard off, measure host time when GPU queue is deep (no waits)
