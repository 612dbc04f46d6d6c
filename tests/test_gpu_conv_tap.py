"""The halo-patch 3x3 kernel (csrc/conv_tap_s16.hip) through `ammc_conv_gemm_s16`, against an fp64 convolution of
the same S16-rounded operands: shapes that the dispatcher sends to it (>= 192 tiles, Cin % 32 == 0, W % 32 == 0,
H % 8 == 0, N = 64 | 128k) with folded-BN scale/shift, ReLU, an S16 residual, and a channel-sliced input (the
concat buffers of the decoder).  Tolerance 2e-6 of max|ref| (fp32-equivalent arithmetic, fp32 accumulation)."""
import os
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from ammcnet_aaai2021_amd import _lib, synthetic as S
from ammcnet_aaai2021_amd._lib import ACT_NONE, ACT_RELU, AmmcConvDesc
from ammcnet_aaai2021_amd.engine import Act, _ptr, s16_variant

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _s16_act(x: torch.Tensor, c_total=None, c_off=0) -> Act:
    """NCHW fp32 (device) -> halo-1 S16 activation occupying channels [c_off, c_off + C) of a wider buffer"""
    lib = _lib.load()
    B, Cc, H, W = x.shape
    c_total = c_total or Cc
    a = Act(torch.zeros(B, H + 2, W + 2, c_total, device=DEV), B, H, W, Cc, c_off, 1)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ammc_nchw_to_s16_f32(_ptr(x), B, Cc, H, W, a.pix0(), *a.strides, Cc, s), "nchw_to_s16")
    return a


def _s16_read(a: Act) -> torch.Tensor:
    lib = _lib.load()
    y = torch.empty(a.B, a.c, a.H, a.W, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ammc_s16_to_nchw_f32(a.pix0(), *a.strides, a.B, a.c, a.H, a.W, _ptr(y), s), "s16_to_nchw")
    return y


# <WGM, WGN, TM, TN, AS, MF>; every variant exists for both MFMA shapes (MF: 1 = 16x16x32, the default; 0 = 32x32x16)
V64, V128, V8W = "conv_tap_s16<4, 1, 2, 2, 1, %d>", "conv_tap_s16<4, 1, 2, 4, 1, %d>", "conv_tap_s16<4, 2, 2, 2, 2, %d>"


# what the DEFAULT dispatch (s16_mf = -1) makes of a forced-shape label: the k-half-major pipelines (KH, a seventh template
# argument) for 64 filters and for 128-filter layers of at least 1024 tiles, the 8-wave 16x16x32 form for the rest
def default_label(variant, tiles):
    if variant == V64:
        return "conv_tap_s16<4, 1, 2, 2, 1, 0, 1>"
    if variant == V128:
        return "conv_tap_s16<4, 1, 2, 4, 1, 0, 1>" if tiles >= 1024 else V8W % 1
    return variant % 1 if "%d" in variant else variant


def expected_label(variant, mf, d):
    if mf >= 0:
        return variant % mf if "%d" in variant else variant
    tiles = d.batch * (d.height // 8) * (d.width // 32) * (1 if d.n <= 64 else d.n // 128)
    return default_label(variant, tiles)


@pytest.fixture(params=[-1, 1, 0], ids=["default", "mfma16x16x32", "mfma32x32x16"])
def mf(request):
    lib = _lib.load()
    _lib.check(lib.ammc_set_option(b"s16_mf", request.param), "set_option")
    yield request.param
    _lib.check(lib.ammc_set_option(b"s16_mf", -1), "set_option")          # back to the per-variant default


@pytest.mark.parametrize("B,H,W,cin,n,relu,res,sliced,variant", [
    (12, 64, 64, 64, 64, True, False, False, V64),        # N = 64: 4 waves x (2 rows x 64 filters)
    (6, 64, 64, 32, 256, False, False, False, V8W),       # 192 tiles: the 8-wave variant, two N tiles, one channel block
    (6, 128, 64, 128, 128, True, True, False, V8W),       # four channel blocks, residual epilogue
    (48, 32, 32, 64, 128, True, False, True, V8W),        # one tile per image, input = slice of a 128-channel buffer
    (1, 256, 256, 64, 64, True, False, False, V64),       # the 256x256 level of the network
    # the benchmark's dominant variant (>= 512 tiles, 128-filter tiles, one accumulator set, two workgroups per CU)
    (16, 64, 64, 32, 256, False, False, False, V128),     # 512 tiles, two N tiles
    (8, 128, 128, 128, 128, True, True, False, V128),     # 512 tiles, residual
    (32, 64, 64, 128, 128, True, False, True, V128),      # sliced input (the decoder's concat buffer)
    (16, 128, 128, 64, 128, True, False, False, V128),    # down1.0 of the benchmark (1024 tiles: the KH instance by default)
    (16, 128, 128, 128, 128, True, True, False, V128),    # 1024 tiles, four channel blocks, residual
    (64, 64, 64, 128, 128, True, False, True, V128),      # 1024 tiles, sliced input
])
def test_conv_tap_s16_vs_fp64(B, H, W, cin, n, relu, res, sliced, variant, mf):
    lib = _lib.load()
    tag = f"tap-{B}-{H}-{W}-{cin}-{n}"
    x = S.hashed_uniform(tag + "x", (B, cin, H, W)).to(DEV)
    w = (S.hashed_uniform(tag + "w", (n, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    scale = S.hashed_uniform(tag + "s", (n,), 0.7, 1.3).to(DEV)
    shift = S.hashed_uniform(tag + "b", (n,), -0.2, 0.2).to(DEV)
    xa = _s16_act(x, 2 * cin if sliced else cin, cin if sliced else 0)
    ra = _s16_act(S.hashed_uniform(tag + "r", (B, n, H, W)).to(DEV)) if res else None
    ya = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
    wp = torch.empty(n, 9 * cin, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ammc_pack_conv_weight_f32(_ptr(w), n, cin, 3, cin, _ptr(wp), s), "pack")
    ws = torch.empty_like(wp)
    _lib.check(lib.ammc_split_rows_f32(_ptr(wp), wp.numel(), _ptr(ws), s), "split")
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    d = AmmcConvDesc()
    d.x, d.w, d.y = xa.tap0(), _ptr(ws), ya.pix0()
    d.scale, d.shift = _ptr(scale), _ptr(shift)
    d.res = ra.pix0() if res else None
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup = B, H, W, cin, 9, n, 1, n
    d.act = ACT_RELU if relu else ACT_NONE
    d.x_bs, d.x_rs, d.x_ps = xa.strides
    d.y_bs, d.y_rs, d.y_ps = ya.strides
    if res:
        d.r_bs, d.r_rs, d.r_ps = ra.strides
    d.overflow_flag = flag.data_ptr()
    pa = None
    if not res:                                        # second output: the 2x2 max-pool of y (the `down` blocks)
        pa = Act(torch.zeros(B, H // 2 + 2, W // 2 + 2, n, device=DEV), B, H // 2, W // 2, n, 0, 1)
        d.pool_y = pa.pix0()
        d.pool_bs, d.pool_rs, d.pool_ps = pa.strides
    if not os.environ.get("AMMC_TAP_KH"):          # (experiments force other instances through the environment)
        assert s16_variant(d) == expected_label(variant, mf, d)                               # the library's own dispatch: this case reaches that kernel
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv_gemm_s16")
    got = _s16_read(ya).double().cpu()
    # reference on the operands as the kernel sees them (S16 round trip of x, w and the residual)
    xr = _s16_read(xa).double().cpu()
    wr = torch.empty_like(wp)
    wsa = Act(ws.view(1, 1, n, 9 * cin), 1, 1, n, 9 * cin, 0, 0)
    wr = _s16_read(wsa).view(9 * cin, n).t().reshape(n, 9, cin).permute(0, 2, 1).reshape(n, cin, 3, 3).double().cpu()
    want = F.conv2d(xr, wr, padding=1) * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)
    if relu:
        want = want.clamp_min(0)
    if res:
        want = want + _s16_read(ra).double().cpu()
    err = float((got - want).abs().max() / want.abs().max())
    assert err <= 2e-6, err
    assert int(flag.item()) == 0
    assert float(ya.buf[:, 0].abs().max()) == 0.0 and float(ya.buf[:, :, 0].abs().max()) == 0.0     # halo untouched
    if pa is not None:
        # S16 rounding is monotone, so the pooled output is EXACTLY the max-pool of the stored (fp64-checked) output
        assert torch.equal(_s16_read(pa), F.max_pool2d(_s16_read(ya), 2))
        assert float(pa.buf[:, 0].abs().max()) == 0.0 and float(pa.buf[:, :, 0].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,W,cin,n,variant", [(12, 64, 64, 64, 64, V64), (12, 64, 64, 64, 128, V8W), (16, 64, 64, 32, 256, V128),
                                                 (48, 32, 32, 64, 128, V8W)])
def test_conv_tap_fp32_output_with_fp32_residual(B, H, W, cin, n, variant, mf):
    """the form the training path uses (train.py `_Ops.conv_s16`): S16 operands, fp32 NHWC output, per-column scale
    (the undo of the gradient rescaling), fp32 residual - every conv_tap_s16 variant (4-wave 64x64 and 64x128 with
    one accumulator set, 8-wave with two)"""
    lib = _lib.load()
    tag = f"tapf-{B}-{H}-{W}-{cin}-{n}"
    x = S.hashed_uniform(tag + "x", (B, cin, H, W)).to(DEV)
    w = (S.hashed_uniform(tag + "w", (n, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    scale = torch.full((n,), 0.25, device=DEV)
    xa = _s16_act(x)
    res = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
    res.interior().copy_(S.hashed_uniform(tag + "r", (B, H, W, n)).to(DEV))
    ya = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
    wp = torch.empty(n, 9 * cin, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ammc_pack_conv_weight_f32(_ptr(w), n, cin, 3, cin, _ptr(wp), s), "pack")
    ws = torch.empty_like(wp)
    _lib.check(lib.ammc_split_rows_f32(_ptr(wp), wp.numel(), _ptr(ws), s), "split")
    d = AmmcConvDesc()
    d.x, d.w, d.y, d.scale, d.res = xa.tap0(), _ptr(ws), ya.pix0(), _ptr(scale), res.pix0()
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act, d.y_f32 = B, H, W, cin, 9, n, 1, n, ACT_NONE, 1
    d.x_bs, d.x_rs, d.x_ps = xa.strides
    d.y_bs, d.y_rs, d.y_ps = ya.strides
    d.r_bs, d.r_rs, d.r_ps = res.strides
    if not os.environ.get("AMMC_TAP_KH"):
        assert s16_variant(d) == expected_label(variant, mf, d)
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv_gemm_s16")
    got = ya.interior().permute(0, 3, 1, 2).double().cpu()
    wsa = Act(ws.view(1, 1, n, 9 * cin), 1, 1, n, 9 * cin, 0, 0)
    wr = _s16_read(wsa).view(9 * cin, n).t().reshape(n, 9, cin).permute(0, 2, 1).reshape(n, cin, 3, 3).double().cpu()
    want = F.conv2d(_s16_read(xa).double().cpu(), wr, padding=1) * 0.25 + res.interior().permute(0, 3, 1, 2).double().cpu()
    err = float((got - want).abs().max() / want.abs().max())
    assert err <= 2e-6, err
    assert float(ya.buf[:, 0].abs().max()) == 0.0 and float(ya.buf[:, :, 0].abs().max()) == 0.0


@pytest.fixture(params=[1, 0], ids=["stream", "tap"])
def outc_stream(request):
    """the output layer on its streaming kernel (default) and on the halo-patch kernel"""
    lib = _lib.load()
    _lib.check(lib.ammc_set_option(b"outc_stream", request.param), "set_option")
    yield request.param
    _lib.check(lib.ammc_set_option(b"outc_stream", 1), "set_option")


def _outc_case(lib, B, H, W, cout, score=True, cin=64):
    s = torch.cuda.current_stream().cuda_stream
    tag = f"outc-{B}-{H}-{W}-{cout}"
    x = S.hashed_uniform(tag + "x", (B, cin, H, W)).to(DEV)
    w = torch.zeros(32, cin, 3, 3)
    w[:cout] = S.hashed_uniform(tag + "w", (cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5
    w = w.to(DEV)
    bias = torch.zeros(32, device=DEV)
    bias[:cout] = S.hashed_uniform(tag + "b", (cout,), -0.2, 0.2).to(DEV)
    target = S.hashed_uniform(tag + "t", (B, cout, H, W)).to(DEV)
    xa = _s16_act(x)
    wp = torch.empty(32, 9 * cin, device=DEV)
    _lib.check(lib.ammc_pack_conv_weight_f32(_ptr(w), 32, cin, 3, cin, _ptr(wp), s), "pack")
    ws = torch.empty_like(wp)
    _lib.check(lib.ammc_split_rows_f32(_ptr(wp), wp.numel(), _ptr(ws), s), "split")
    y = torch.empty(B, cout, H, W, device=DEV)
    sq = torch.zeros(B, device=DEV)
    d = AmmcConvDesc()
    d.x, d.w, d.y, d.shift = xa.tap0(), _ptr(ws), _ptr(y), _ptr(bias)
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act, d.n_store, d.y_f32 = B, H, W, cin, 9, 32, 1, 32, 2, cout, 1
    d.x_bs, d.x_rs, d.x_ps = xa.strides
    d.y_bs, d.y_rs, d.y_ps, d.y_cs = cout * H * W, W, 1, H * W
    if score:
        d.sq_target, d.sq_acc = _ptr(target), _ptr(sq)
    keep = (x, w, bias, target, xa, wp, ws)
    return d, y, sq, keep


@pytest.mark.parametrize("B,H,W,cout,cin", [(8, 64, 96, 3, 64), (16, 256, 256, 3, 64), (7, 80, 96, 2, 64), (5, 128, 160, 3, 64),
                                             (5, 128, 160, 3, 32), (8, 64, 96, 1, 32)])
def test_output_layer_streaming_kernel_vs_halo_patch_kernel(B, H, W, cout, cin):
    """conv_outc_s16 (persistent, three patch stages, two accumulator sets) against conv_tap_s16<.., MF = 1> (one
    accumulator set) on the same operands: frames equal to fp32 rounding of the sums (the benchmark's size; 210 tiles =
    fewer than the grid; 400 tiles = workgroups with one and with two tiles; 32 input channels = one unit per tile),
    squared errors to the order of their fp32 sums; without a target the frames are the same bits and nothing is written to sq_acc"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    try:
        for stream in (1, 0):
            _lib.check(lib.ammc_set_option(b"outc_stream", stream), "set_option")
            _lib.check(lib.ammc_set_option(b"s16_mf", 1), "set_option")
            d, y, sq, keep = _outc_case(lib, B, H, W, cout, cin=cin)
            assert s16_variant(d) == ("conv_outc_s16" if stream else "conv_tap_s16<4, 1, 2, 1, 1, 1>")
            _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "outc")
            out[stream] = (y.clone(), sq.clone())
        d, y, sq, keep = _outc_case(lib, B, H, W, cout, score=False, cin=cin)
        _lib.check(lib.ammc_set_option(b"outc_stream", 1), "set_option")
        _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "outc")
        assert torch.equal(y, out[1][0]) and float(sq.abs().max()) == 0.0
    finally:
        _lib.check(lib.ammc_set_option(b"outc_stream", 1), "set_option")
        _lib.check(lib.ammc_set_option(b"s16_mf", -1), "set_option")
    assert float((out[1][0] - out[0][0]).abs().max()) <= 1e-6
    assert float(((out[1][1] - out[0][1]).abs() / out[0][1]).max()) <= 1e-5


@pytest.mark.parametrize("B,H,W,cout", [(8, 64, 96, 3), (3, 128, 128, 2)])
def test_output_layer_vs_fp64(B, H, W, cout, mf, outc_stream):
    """`outc` + tanh (reference unet.py:920, 998-1007) as the benchmark runs it: 32-filter tile with `n_store` real
    filters, fp32 NCHW frames, bias, tanh, and the fused per-sample squared error of `psnr_error` (utils.py:141-148) -
    the streaming kernel (conv_outc_s16.hip) and both instances of the halo-patch kernel (MF 1: 4 waves, 16-filter MFMA
    tile, three workgroups per CU; MF 0: the 8-wave 32x32x16 form) against fp64 on the S16-rounded operands"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    cin = 64
    d, y, sq, (x, w, bias, target, xa, wp, ws) = _outc_case(lib, B, H, W, cout)
    assert s16_variant(d) == ("conv_outc_s16" if outc_stream else
                              ("conv_tap_s16<4, 1, 2, 1, 1, 1>" if mf else "conv_tap_s16<8, 1, 1, 1, 1, 0>"))
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "outc")
    wsa = Act(ws.view(1, 1, 32, 9 * cin), 1, 1, 32, 9 * cin, 0, 0)
    wr = _s16_read(wsa).view(9 * cin, 32).t().reshape(32, 9, cin).permute(0, 2, 1).reshape(32, cin, 3, 3).double().cpu()
    want = torch.tanh(F.conv2d(_s16_read(xa).double().cpu(), wr[:cout], padding=1) + bias[:cout].double().cpu().view(1, -1, 1, 1))
    assert float((y.double().cpu() - want).abs().max()) <= 2e-6
    want_sq = ((target.double().cpu() - want) * 0.5).pow(2).sum(dim=(1, 2, 3))
    assert float(((sq.double().cpu() - want_sq).abs() / want_sq).max()) <= 2e-5


@pytest.mark.parametrize("case", ["enc", "dec"])
def test_conv1x1_s16_vs_fp64(case):
    """the memory block's 1x1 convs on `ammc_conv_gemm_s16` (unet.py:321-330, 386): `enc` 512 -> 64 with bias and an fp32
    output (what the memory kernel reads), `dec` 128 -> 512 with bias and the S16 residual (`out += x`)"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    B, H, W = 16, 32, 32
    cin, n = (512, 64) if case == "enc" else (128, 512)
    x = S.hashed_uniform(case + "x", (B, cin, H, W)).to(DEV)
    w = (S.hashed_uniform(case + "w", (n, cin, 1, 1)) * (1.0 / cin) ** 0.5).to(DEV)
    bias = S.hashed_uniform(case + "b", (n,), -0.2, 0.2).to(DEV)
    xa = _s16_act(x)
    wp = torch.empty(n, cin, device=DEV)
    _lib.check(lib.ammc_pack_conv_weight_f32(_ptr(w), n, cin, 1, cin, _ptr(wp), s), "pack")
    ws = torch.empty_like(wp)
    _lib.check(lib.ammc_split_rows_f32(_ptr(wp), wp.numel(), _ptr(ws), s), "split")
    d = AmmcConvDesc()
    d.x, d.w, d.shift = xa.pix0(), _ptr(ws), _ptr(bias)
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act = B, H, W, cin, 1, n, 1, n, ACT_NONE
    d.x_bs, d.x_rs, d.x_ps = xa.strides
    wsa = Act(ws.view(1, 1, n, cin), 1, 1, n, cin, 0, 0)
    wr = _s16_read(wsa).view(cin, n).t().reshape(n, cin, 1, 1).double().cpu()
    want = F.conv2d(_s16_read(xa).double().cpu(), wr) + bias.double().cpu().view(1, -1, 1, 1)
    if case == "enc":
        ya = Act(torch.zeros(B, H, W, n, device=DEV), B, H, W, n, 0, 0)
        d.y, d.y_f32 = ya.pix0(), 1
        d.y_bs, d.y_rs, d.y_ps = ya.strides
        assert s16_variant(d) == "conv_gemm_s16<128x64>"
        _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "enc")
        got = ya.interior().permute(0, 3, 1, 2).double().cpu()
    else:
        res = S.hashed_uniform(case + "r", (B, n, H, W)).to(DEV)
        ra = _s16_act(res)
        ya = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
        d.y, d.res = ya.pix0(), ra.pix0()
        d.y_bs, d.y_rs, d.y_ps = ya.strides
        d.r_bs, d.r_rs, d.r_ps = ra.strides
        assert s16_variant(d) == "conv_gemm_s16<128x128>"
        _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "dec")
        got = _s16_read(ya).double().cpu()
        want = want + _s16_read(ra).double().cpu()
    assert float((got - want).abs().max() / want.abs().max()) <= 2e-6
