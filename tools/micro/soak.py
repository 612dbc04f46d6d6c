"""race screen for the DMA-pipelined kernels: N forwards of the same batch must give the same bits (no atomics on the
frame path); also at other batch sizes (different dispatch mixes) and under memory pressure from a concurrent copy stream"""
import sys
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
net = A.get_twostream((12, 6), (3, 2), 64, 2000, 2)
net.load_state_dict(S.make_twostream_state(n_embed=2000))
net = net.cuda().eval()
side = torch.cuda.Stream()
big = torch.empty(256 << 20, device="cuda", dtype=torch.float32)          # 1 GB: a copy stream hammering HBM beside the forwards
for B in (16, 5, 32):
    rgb, op, _, _ = (t.cuda() for t in S.make_clips(B, 256, 256, tag=f"soak{B}"))
    ref, bad = None, 0
    with torch.no_grad():
        for i in range(N):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    big[: 128 << 20].copy_(big[128 << 20:], non_blocking=True)
            out = net(rgb, op)
            got = (out[0], out[1], out[3][0], out[3][1])
            if ref is None:
                ref = [t.clone() for t in got]
            else:
                bad += int(not all(torch.equal(a, b) for a, b in zip(got, ref)))
    torch.cuda.synchronize()
    print(f"batch {B}: {N} forwards, {bad} differ from the first, fallbacks {getattr(net, 's16_fallbacks', 0)}", flush=True)
