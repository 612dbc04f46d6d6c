"""S16 3x3 conv layers in isolation (HIP events, RANDOM operands: the chip's clock under load depends on the data):

    python tools/conv_bench.py B H W CIN N [reps]          one shape
    python tools/conv_bench.py --net [B] [reps]            the 3x3 layer shapes of the 256x256 network at batch B (16)

env: AMMC_S16_TAP=0 (GEMM kernel), AMMC_S16_DBG=2 (no MFMA) / 3 (no DMA in the loop; tap kernel only, -DAMMC_TAP_DEBUG),
     AMMC_LIB=path of another build of the library (A/Bs inside one gpurun call)"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import _lib

lib = _lib.load()
dev = "cuda:0"
NET = [(256, 256, 64, 64), (128, 128, 64, 128), (128, 128, 128, 128), (64, 64, 128, 256), (64, 64, 256, 256),
       (32, 32, 256, 512), (32, 32, 512, 512), (64, 64, 512, 256), (128, 128, 256, 128), (256, 256, 128, 64)]


from tools.conv_bench_lib import make_desc


def bench(B, H, W, cin, n, reps):
    d, keep = make_desc(B, H, W, cin, n)
    s = torch.cuda.current_stream().cuda_stream
    label = C.create_string_buffer(96)
    lib.ammc_conv_gemm_s16_variant(C.byref(d), label, 96)
    for _ in range(3):
        _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.ammc_conv_gemm_s16(C.byref(d), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * B * H * W * 9 * cin * n
    print(f"B={B} {H}x{W} {cin:3d}->{n:3d} {label.value.decode():34s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF algorithmic  "
          f"({3 * fl / us / 1e6 / 2500 * 100:4.1f}% of the f16 MFMA issue peak)", flush=True)
    return us


if sys.argv[1] == "--net":
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    tot = 0.0
    for (H, W, cin, n) in NET:
        tot += bench(B, H, W, cin, n, reps)
    print(f"sum {tot:8.1f} us")
else:
    B, H, W, cin, n = (int(v) for v in sys.argv[1:6])
    bench(B, H, W, cin, n, int(sys.argv[6]) if len(sys.argv) > 6 else 20)
