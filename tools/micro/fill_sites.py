"""which python call sites launch fill kernels in one training step (zero_, zeros, zeros_like, fill_, full)"""
import collections, sys, traceback
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import harness, synthetic as S
net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
net.load_state_dict(S.make_twostream_state())
net = net.cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
B = 4
rgb_x, op_x, rgb_t, op_t = (t.cuda() for t in S.make_clips(B, 256, 256, tag="fills"))
rgb = torch.cat([rgb_x.view(B, 4, 3, 256, 256), rgb_t[:, None]], 1)
op = torch.cat([op_x.view(B, 3, 2, 256, 256), op_t[:, None]], 1)
for _ in range(2):
    harness.train_step(net, opt, rgb, op)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    harness.train_step(net, opt, rgb, op)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::full", "aten::ones_like", "aten::ones"):
        st = [f for f in (ev.stack or []) if "ammcnet" in f or "harness" in f or "optim" in f]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
ka = prof.key_averages()
for e in sorted(ka, key=lambda e: -e.count)[:12]:
    print(e.key[:70], e.count)
