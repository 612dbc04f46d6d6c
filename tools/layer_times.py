"""per-launch timing of one forward (HIP events): python tools/layer_times.py [precision] [batch]"""
import sys
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
prec = sys.argv[1] if len(sys.argv) > 1 else "s16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
net = A.get_twostream((12, 6), (3, 2), 64, 2000, 2)
net.load_state_dict(S.make_twostream_state(n_embed=2000))
net = net.cuda().eval(); net.precision = prec
rgb, op, _, _ = S.make_clips(B, 256, 256, tag="lt")
rgb, op = rgb.cuda(), op.cuda()
for _ in range(3): net(rgb, op)
eng = net._engine; eng._timed = True
acc = {}
for _ in range(3):
    net(rgb, op)
    for i, (m, ms) in enumerate(eng.timings):
        a = acc.setdefault(i, [m, 0.0]); a[1] += ms / 3
tot = sum(v[1] for v in acc.values())
print(f"total {tot:.3f} ms")
for i, (m, ms) in acc.items():
    tf = m['flops'] / ms / 1e9 if m['flops'] else 0
    print(f"{i:3d} {m['name']:22s} {m['kernel']:26s} {ms*1e3:9.1f} us  {tf:7.1f} TF  {m['bytes']/ms/1e6 if m['bytes'] else 0:8.0f} GB/s")
