"""one S16 3x3 conv layer in isolation (HIP events):  python tools/conv_bench.py B H W CIN N [reps]
env: AMMC_S16_TAP=0 (GEMM kernel), AMMC_S16_DBG=2 (no MFMA) / 3 (no DMA in the loop; tap kernel only)"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import _lib
from ammcnet_aaai2021_amd._lib import ACT_RELU, AmmcConvDesc
from ammcnet_aaai2021_amd.engine import Act, _ptr
B, H, W, cin, n = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
lib = _lib.load()
dev = "cuda:0"
xa = Act(torch.zeros(B, H + 2, W + 2, cin, device=dev), B, H, W, cin, 0, 1)
xa.interior().copy_(torch.randn(B, H, W, cin, device=dev) * 0.0 + 1e-3)      # harmless S16 bit patterns
ya = Act(torch.zeros(B, H + 2, W + 2, n, device=dev), B, H, W, n, 0, 1)
ws = torch.zeros(n, 9 * cin, device=dev)
scale = torch.ones(n, device=dev); shift = torch.zeros(n, device=dev)
d = AmmcConvDesc()
d.x, d.w, d.y, d.scale, d.shift = xa.tap0(), _ptr(ws), ya.pix0(), _ptr(scale), _ptr(shift)
d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act = B, H, W, cin, 9, n, 1, n, ACT_RELU
d.x_bs, d.x_rs, d.x_ps = xa.strides
d.y_bs, d.y_rs, d.y_ps = ya.strides
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    lib.ammc_conv_gemm_s16(C.byref(d), s)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * B * H * W * 9 * cin * n
print(f"B={B} {H}x{W} {cin}->{n}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF algorithmic  ({3 * fl / us / 1e6 / 2500 * 100:4.1f}% of the f16 MFMA issue peak)")
