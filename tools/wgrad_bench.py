"""time one 3x3 S16 weight-gradient layer:  python tools/wgrad_bench.py B H W cin n [reps]
env: AMMC_WGRAD_TAP=2 (four-product halo-patch kernels), =0 (im2col kernel)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ammcnet_aaai2021_amd import _lib
from ammcnet_aaai2021_amd._lib import AmmcWgradDesc
from ammcnet_aaai2021_amd.engine import Act, _ptr

B, H, W, cin, n = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = "cuda:0"
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
A32 = Act(torch.zeros(B, H + 2, W + 2, cin, device=dev), B, H, W, cin, 0, 1)
A32.interior().copy_(torch.rand(B, H, W, cin, device=dev, generator=g) * 2 - 1)
G32 = Act(torch.zeros(B, H + 2, W + 2, n, device=dev), B, H, W, n, 0, 1)
G32.interior().copy_(torch.rand(B, H, W, n, device=dev, generator=g) * 2 - 1)
A16 = Act(torch.empty_like(A32.buf), B, H, W, cin, 0, 1)
G16 = Act(torch.empty_like(G32.buf), B, H, W, n, 0, 1)
_lib.check(lib.ammc_split_rows_f32(_ptr(A32.buf), A32.buf.numel(), _ptr(A16.buf), s), "split a")
_lib.check(lib.ammc_split_rows_f32(_ptr(G32.buf), G32.buf.numel(), _ptr(G16.buf), s), "split g")
kpad = (9 * cin + 31) // 32 * 32
dwp = torch.zeros(max(n, 32), kpad, device=dev)
zeros = torch.zeros(1024, device=dev)
d = AmmcWgradDesc()
d.g, d.a, d.dw, d.zeros = G16.pix0(), A16.tap0(), _ptr(dwp), _ptr(zeros)
d.batch, d.height, d.width, d.n, d.cin, d.ntaps, d.a_step = B, H, W, n, cin, 9, 1
d.g_bs, d.g_rs, d.g_ps = G16.strides
d.a_bs, d.a_rs, d.a_ps = A16.strides
# the training path's form (train.py: _Ops.wgrad_s16): partials as slabs + the fixed-order reduce into OIHW; WGRAD_SLABS=0: atomics
need = int(lib.ammc_conv_wgrad_s16_slab_floats(C.byref(d))) if os.environ.get("WGRAD_SLABS", "1") != "0" else 0
if need:
    slabs = torch.empty(need, device=dev)
    out = torch.empty(n, cin, 3, 3, device=dev)
    call = lambda: lib.ammc_conv_wgrad_s16_slabs(C.byref(d), None, _ptr(slabs), need, _ptr(out), n, cin, s)
else:
    call = lambda: lib.ammc_conv_wgrad_s16(C.byref(d), None, s)
for _ in range(3):
    _lib.check(call(), "wgrad")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
fl = 2.0 * B * H * W * n * cin * 9
print(f"B={B} {H}x{W} {cin}->{n} ({'slabs + reduce' if need else 'atomics'}): {us:8.1f} us  {fl / us / 1e6:7.1f} TF algorithmic  ({3 * fl / us / 1e6 / 2500 * 100:.1f}% of the f16 MFMA issue peak with 3 MFMAs)")
