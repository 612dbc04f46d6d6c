"""Winograd F(2x2, 3x3) on the Cin >= 256 layers, measured UNFUSED:

    input transform (S16 -> sixteen S16 planes)  ->  sixteen 1x1 S16 GEMMs (ammc_conv_gemm_s16, ntaps 1, fp32 out, spread
    over parallel HIP streams)  ->  output transform (+ BatchNorm scale / shift, ReLU, S16 store)

against the direct halo-patch kernel the model runs on the same layer: microseconds and board power of each phase on
random post-ReLU operands, and the accuracy of both against an fp64 convolution.

    python tools/micro/wino_proto.py [seconds per case]        (B=16: 512->512 @32x32, 256->256 @64x64)

What to read from it: the GEMM phase is what a FUSED Winograd kernel's MFMA stream would cost at best (2.25x fewer MFMAs,
16/9 more filter bytes, sixteen accumulator planes); the transform phases are what fusion would have to hide."""
import ctypes as C
import glob
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import _lib
from ammcnet_aaai2021_amd._lib import ACT_RELU, AmmcConvDesc
from ammcnet_aaai2021_amd.engine import Act, _ptr

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libwino_proto.so")
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
dev = "cuda:0"


def build():
    src = os.path.join(HERE, "wino_proto.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
                               "-shared", src, "-o", SO])
    return C.CDLL(SO)


def hwmon():
    pr = torch.cuda.get_device_properties(0)
    want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        bdf = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
        if bdf.lower().startswith(want) and os.path.exists(os.path.join(d, "power1_input")):
            return d
    return None


def rd(path):
    try:
        with open(path) as fp:
            return float(fp.read().strip())
    except Exception:
        return None


HW = None


def run_case(name, launch, flops, sync_all):
    """`launch()` enqueues one unit of work; keeps the queues full for SEC seconds while a thread samples power / clock"""
    for _ in range(5):
        launch()
    sync_all()
    stop, samples = threading.Event(), []

    def loop():
        while not stop.is_set():
            if HW:
                samples.append((rd(os.path.join(HW, "power1_input")), rd(os.path.join(HW, "freq1_input"))))
            time.sleep(0.02)
    th = threading.Thread(target=loop)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < SEC:
        for _ in range(20):
            launch()
        n += 20
        sync_all()
    us = (time.time() - t0) / n * 1e6
    stop.set()
    th.join()
    mid = [s for s in samples[len(samples) // 3:] if s[0]]
    pw = sum(p for p, _ in mid) / max(len(mid), 1) / 1e6
    fq = sum(f for _, f in mid) / max(len(mid), 1) / 1e6
    print(f"  {name:58s} {us:8.1f} us  {pw:7.1f} W  {fq:7.1f} MHz  {pw * us * 1e-6:7.4f} J" +
          (f"  {flops / us / 1e6:7.1f} TF alg" if flops else ""), flush=True)
    return us, pw


def s16_of(lib, t):
    out = torch.empty_like(t)
    _lib.check(lib.ammc_split_rows_f32(_ptr(t), t.numel(), _ptr(out), torch.cuda.current_stream().cuda_stream), "split")
    return out


def s16_decode(buf, c):
    """S16 buffer [..., c] (fp32-sized storage) -> fp32 values"""
    h = buf.contiguous().view(torch.float16).reshape(*buf.shape[:-1], c // 8, 2, 8).float()
    return (h[..., 0, :] + h[..., 1, :] / 2048.0).reshape(*buf.shape[:-1], c)


def layer(lib, wl, B, H, W, cin, n, streams):
    print(f"--- {cin} -> {n} @ {H}x{W}, batch {B}: {2.0 * B * H * W * 9 * cin * n / 1e9:.1f} GFLOP direct, "
          f"{2.0 * B * H * W * 4 * cin * n / 1e9:.1f} GFLOP in the Winograd GEMMs", flush=True)
    g = torch.Generator(device=dev).manual_seed(7)
    s0 = torch.cuda.current_stream().cuda_stream
    x32 = torch.zeros(B, H + 2, W + 2, cin, device=dev)
    x32[:, 1:-1, 1:-1].copy_(torch.randn(B, H, W, cin, device=dev, generator=g).clamp_min(0))      # post-ReLU statistics
    w = torch.randn(n, 3, 3, cin, device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5              # [n][r][s][c]
    scale = torch.rand(n, device=dev, generator=g) + 0.5
    shift = torch.randn(n, device=dev, generator=g) * 0.1
    xa = Act(s16_of(lib, x32), B, H, W, cin, 0, 1)
    # ---- direct: the product's own kernel
    ya = Act(torch.zeros(B, H + 2, W + 2, n, device=dev), B, H, W, n, 0, 1)
    wp = s16_of(lib, w.reshape(n, 9 * cin).contiguous())
    d = AmmcConvDesc()
    d.x, d.w, d.y, d.scale, d.shift = xa.tap0(), _ptr(wp), ya.pix0(), _ptr(scale), _ptr(shift)
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act = B, H, W, cin, 9, n, 1, n, ACT_RELU
    d.x_bs, d.x_rs, d.x_ps = xa.strides
    d.y_bs, d.y_rs, d.y_ps = ya.strides
    label = C.create_string_buffer(96)
    lib.ammc_conv_gemm_s16_variant(C.byref(d), label, 96)
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s0), "direct")
    # ---- Winograd
    T = B * (H // 2) * (W // 2)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], device=dev, dtype=torch.float64)
    U = torch.einsum("ar,nrsc,bs->abnc", G, w.double(), G).reshape(16, n, cin).float().contiguous()   # U[xi][n][c], summed in double
    Us = s16_of(lib, U)
    V = torch.empty(16, T, cin, device=dev)
    M = torch.empty(16, T, n, device=dev)
    yw = Act(torch.zeros(B, H + 2, W + 2, n, device=dev), B, H, W, n, 0, 1)
    descs = []
    for xi in range(16):
        q = AmmcConvDesc()
        q.x, q.w, q.y = _ptr(V, xi * T * cin), _ptr(Us, xi * n * cin), _ptr(M, xi * T * n)
        q.batch, q.height, q.width, q.cin, q.ntaps, q.n, q.up, q.cgroup, q.act = 1, 1, T, cin, 1, n, 1, n, 0
        q.x_bs, q.x_rs, q.x_ps = T * cin, T * cin, cin
        q.y_bs, q.y_rs, q.y_ps = T * n, T * n, n
        q.y_f32 = 1
        descs.append(q)
    lib.ammc_conv_gemm_s16_variant(C.byref(descs[0]), label2 := C.create_string_buffer(96), 96)

    def t_in(s=s0):
        wl.wino_input_transform(C.c_void_p(xa.tap0()), *[C.c_int64(v) for v in xa.strides], B, H, W, cin, C.c_void_p(_ptr(V)), C.c_void_p(s))

    def t_gemm():
        for xi, q in enumerate(descs):
            lib.ammc_conv_gemm_s16(C.byref(q), streams[xi % len(streams)].cuda_stream)

    def t_out(s=s0):
        wl.wino_output_transform(C.c_void_p(_ptr(M)), B, H, W, n, C.c_void_p(_ptr(scale)), C.c_void_p(_ptr(shift)), 1,
                                 C.c_void_p(yw.pix0()), *[C.c_int64(v) for v in yw.strides], C.c_void_p(s))

    t_in()
    torch.cuda.synchronize()
    t_gemm()
    torch.cuda.synchronize()
    t_out()
    torch.cuda.synchronize()
    # ---- accuracy: both against an fp64 convolution of the SAME S16-rounded operands
    xd = s16_decode(xa.buf, cin).double()
    wd = s16_decode(wp, 9 * cin).double().reshape(n, 3, 3, cin)
    ref = torch.zeros(B, H, W, n, device=dev, dtype=torch.float64)
    for r in range(3):
        for s in range(3):
            ref += torch.einsum("bhwc,nc->bhwn", xd[:, r:r + H, s:s + W], wd[:, r, s])
    ref = (ref * scale.double() + shift.double()).clamp_min(0)
    for name, act in (("direct " + label.value.decode(), ya), ("winograd F(2x2,3x3), fp32 transforms", yw)):
        got = s16_decode(act.buf, n)[:, 1:-1, 1:-1].double()
        print(f"  accuracy {name:50s} max|d| / max|ref| = {float((got - ref).abs().max() / ref.abs().max()):.2e}", flush=True)
    del xd, wd, ref
    # ---- time and power, phase by phase
    sync = torch.cuda.synchronize
    fl = 2.0 * B * H * W * 9 * cin * n
    ud, pd = run_case("direct: " + label.value.decode(), lambda: lib.ammc_conv_gemm_s16(C.byref(d), s0), fl, sync)
    ui, pi = run_case("winograd input transform (S16 -> 16 S16 planes)", t_in, 0, sync)
    ug, pg = run_case(f"winograd 16 x {label2.value.decode()} on {len(streams)} streams", t_gemm, fl, sync)
    uo, po = run_case("winograd output transform (+ BN, ReLU, S16 store)", t_out, 0, sync)
    print(f"  => direct {ud:.1f} us / {pd * ud * 1e-6:.4f} J;  winograd unfused {ui + ug + uo:.1f} us / "
          f"{(pi * ui + pg * ug + po * uo) * 1e-6:.4f} J  (GEMM phase alone {ug:.1f} us / {pg * ug * 1e-6:.4f} J = "
          f"{ug / ud:.2f}x the direct kernel's time, {pg * ug / (pd * ud):.2f}x its energy)", flush=True)


def main():
    global HW
    lib = _lib.load()
    wl = build()
    HW = hwmon()
    print(f"hwmon {HW}  cap {(rd(os.path.join(HW, 'power1_cap')) or 0) / 1e6:.0f} W" if HW else "no hwmon", flush=True)
    for ns in (4, 16):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        for (B, H, W, cin, n) in [(16, 32, 32, 512, 512), (16, 64, 64, 256, 256)]:
            layer(lib, wl, B, H, W, cin, n, streams)


if __name__ == "__main__":
    main()
