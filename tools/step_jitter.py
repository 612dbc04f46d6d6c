"""wall time of blocks of 5 forwards, garbage collector on / off: python tools/step_jitter.py [batch] [blocks]"""
import gc, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
net = A.get_twostream((12, 6), (3, 2), 64, 2000, 2)
net.load_state_dict(S.make_twostream_state(n_embed=2000))
net = net.cuda().eval()
rgb, op, _, _ = S.make_clips(B, 256, 256, tag="jit")
rgb, op = rgb.cuda(), op.cuda()
with torch.no_grad():
    for _ in range(5): net(rgb, op)
    for mode in ("gc on", "gc off", "gc on", "gc off"):
        if mode == "gc off":
            gc.collect(); gc.disable()
        else:
            gc.enable()
        ts = []
        for _ in range(blocks):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): net(rgb, op)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
        ts = np.array(ts)
        print(f"{mode}: median {np.median(ts):.3f} ms  mean {ts.mean():.3f}  max {ts.max():.3f}  blocks > 1.15 x median: {(ts > 1.15 * np.median(ts)).sum()} of {blocks}; gc counts {gc.get_count()}")
gc.enable()
