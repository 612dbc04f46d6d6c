"""The C-ABI entry points added in round 4, each against an fp64 / composition reference through ctypes:
  * `ammc_conv_wgrad_s16` with 4x4 windows at stride 1 | 2 (PixelDiscriminator) and the 2x2 stride-2 window of a
    ConvTranspose's weight gradient (the generalised im2col form of csrc/wgrad_s16.hip);
  * `ammc_conv_gemm_s16` with a 4x4 window, stride 2 and the LeakyReLU epilogue, S16 and fp32 outputs;
  * `ammc_pack_filters_s16` = pack + split of every layer, bit for bit, from one launch;
  * `ammc_chan_sum_absmax_f32` + `ammc_split_scaled_strided_f32` on a channel slice of a wider buffer;
  * `ammc_lrelu_s16`."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ammcnet_aaai2021_amd import _lib, synthetic as S
from ammcnet_aaai2021_amd._lib import ACT_LRELU, AmmcConvDesc, AmmcWgradDesc
from ammcnet_aaai2021_amd.engine import Act, _kpad, _ptr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _s16(lib, t):
    out = torch.empty_like(t)
    _lib.check(lib.ammc_split_rows_f32(_ptr(t), t.numel(), _ptr(out), torch.cuda.current_stream().cuda_stream), "split")
    return out


def _decode(buf, c):
    h = buf.contiguous().view(torch.float16).reshape(*buf.shape[:-1], c // 8, 2, 8).double()
    return (h[..., 0, :] + h[..., 1, :] / 2048.0).reshape(*buf.shape[:-1], c)


@pytest.mark.parametrize("B,H,W,cin,n,stride,gmag", [(2, 20, 28, 8, 128, 2, 1e-6), (3, 17, 13, 128, 256, 2, 1.0),
                                                      (2, 9, 11, 256, 32, 1, 3e-4), (1, 33, 33, 64, 64, 2, 1e-2)])
def test_wgrad_s16_4x4_windows(B, H, W, cin, n, stride, gmag):
    """dW[n][c][r][s] = sum_m G[m][n] A[stride * m + (r, s)][c] over a 2-pixel zero halo (Conv2d(k 4, padding 2))"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    oh, ow = (H + 4 - 4) // stride + 1, (W + 4 - 4) // stride + 1
    tag = f"w4-{B}-{H}-{W}-{cin}-{n}-{stride}"
    a = S.hashed_uniform(tag + "a", (B, H, W, cin)).to(DEV)
    g = (S.hashed_uniform(tag + "g", (B, oh, ow, n)) * gmag).to(DEV)
    A32 = Act(torch.zeros(B, H + 4, W + 4, cin, device=DEV), B, H, W, cin, 0, 2)
    A32.interior().copy_(a)
    G32 = Act(torch.zeros(B, oh + 2, ow + 2, n, device=DEV), B, oh, ow, n, 0, 1)
    G32.interior().copy_(g)
    A16 = _s16(lib, A32.buf)
    G16 = Act(torch.empty_like(G32.buf), B, oh, ow, n, 0, 1)
    amax = torch.zeros(256, dtype=torch.int32, device=DEV)
    inv = torch.empty(8, device=DEV)
    _lib.check(lib.ammc_absmax_bits_f32(_ptr(G32.buf), G32.buf.numel(), amax.data_ptr(), s), "absmax")
    _lib.check(lib.ammc_split_rows_scaled_f32(_ptr(G32.buf), G32.buf.numel(), _ptr(G16.buf), amax.data_ptr(), _ptr(inv), 8, s), "split g")
    kpad = _kpad(16 * cin)
    dwp = torch.zeros(max(n, 32), kpad, device=DEV)
    zeros = torch.zeros(1024, device=DEV)
    d = AmmcWgradDesc()
    d.g, d.a, d.dw, d.zeros = G16.pix0(), _ptr(A16), _ptr(dwp), _ptr(zeros)         # a: the corner of the 2-pixel halo
    d.batch, d.height, d.width, d.n, d.cin, d.ntaps, d.a_step = B, oh, ow, n, cin, 16, stride
    d.g_bs, d.g_rs, d.g_ps = G16.strides
    d.a_bs, d.a_rs, d.a_ps = A32.strides
    _lib.check(lib.ammc_conv_wgrad_s16(C.byref(d), _ptr(inv), s), "wgrad_s16(4x4)")
    dw = torch.empty(n, cin, 4, 4, device=DEV)
    _lib.check(lib.ammc_unpack_conv_wgrad_f32(_ptr(dwp), n, cin, 4, cin, _ptr(dw), s), "unpack")
    w = torch.zeros(n, cin, 4, 4, dtype=torch.float64, requires_grad=True)
    (F.conv2d(a.double().cpu().permute(0, 3, 1, 2), w, stride=stride, padding=2) * g.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    err = float((dw.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
    assert err <= 3e-6, err


@pytest.mark.parametrize("B,h,w,c,gmag", [(2, 8, 16, 64, 1e-5), (3, 5, 7, 128, 1.0)])
def test_wgrad_s16_convtranspose_window(B, h, w, c, gmag):
    """ConvTranspose2d(2c, c, 2, stride 2): dW[ci][co][dy][dx] = sum_m X[m][ci] dY[2m + (dy, dx)][co]; rows = input channels,
    k = (dy * 2 + dx) * c + co (the layout `ammc_unpack_convt_wgrad_f32` reads); the SCALE sits on the `a` operand here"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tag = f"wt-{B}-{h}-{w}-{c}"
    x = S.hashed_uniform(tag + "x", (B, h, w, 2 * c)).to(DEV)
    dy = (S.hashed_uniform(tag + "dy", (B, 2 * h, 2 * w, c)) * gmag).to(DEV)
    X32 = Act(torch.zeros(B, h + 2, w + 2, 2 * c, device=DEV), B, h, w, 2 * c, 0, 1)
    X32.interior().copy_(x)
    # the gradient lives in the upper half of a 2c-channel concat buffer: strided operands
    D32 = Act(torch.zeros(B, 2 * h + 2, 2 * w + 2, 2 * c, device=DEV), B, 2 * h, 2 * w, c, c, 1)
    D32.interior().copy_(dy)
    X16 = Act(_s16(lib, X32.buf), B, h, w, 2 * c, 0, 1)
    D16 = Act(torch.zeros_like(D32.buf), B, 2 * h, 2 * w, c, c, 1)
    amax = torch.zeros(256, dtype=torch.int32, device=DEV)
    inv = torch.empty(16, device=DEV)
    nb = lib.ammc_chan_reduce_blocks(B * 4 * h * w)
    part = torch.zeros(nb * c + 64, device=DEV)
    _lib.check(lib.ammc_chan_sum_absmax_f32(D32.pix0(), *D32.strides, B, 2 * h, 2 * w, c, _ptr(part), amax.data_ptr(), s), "chan_sum_absmax")
    bsum = torch.empty(c, device=DEV)
    _lib.check(lib.ammc_reduce_partials_f32(_ptr(part), nb, c, 1.0, _ptr(bsum), s), "reduce")
    _lib.check(lib.ammc_split_scaled_strided_f32(D32.pix0(), *D32.strides, D16.pix0(), *D16.strides, B, 2 * h, 2 * w, c,
                                                 amax.data_ptr(), _ptr(inv), 16, s), "split_scaled_strided")
    assert float((bsum.double().cpu() - dy.double().cpu().sum((0, 1, 2))).abs().max()) <= 1e-5 * float(dy.abs().sum((0, 1, 2)).max())
    factor = 1.0 / float(inv[0])
    assert 1024.0 <= float(dy.abs().max()) * factor < 2048.0
    got = _decode(D16.buf, 2 * c)[:, 1:-1, 1:-1, c:] / factor                      # the re-encoded slice, scale undone
    assert float((got.cpu() - dy.double().cpu()).abs().max()) <= 3e-7 * float(dy.abs().max())
    assert float(_decode(D16.buf, 2 * c)[..., :c].abs().max()) == 0.0             # the other half of the twin is untouched
    dwp = torch.zeros(2 * c, 4 * c, device=DEV)
    zeros = torch.zeros(1024, device=DEV)
    d = AmmcWgradDesc()
    d.g, d.a, d.dw, d.zeros = X16.pix0(), D16.pix0(), _ptr(dwp), _ptr(zeros)
    d.batch, d.height, d.width, d.n, d.cin, d.ntaps, d.a_step = B, h, w, 2 * c, c, 4, 2
    d.g_bs, d.g_rs, d.g_ps = X16.strides
    d.a_bs, d.a_rs, d.a_ps = D16.strides
    _lib.check(lib.ammc_conv_wgrad_s16(C.byref(d), _ptr(inv), s), "wgrad_s16(convT)")
    dwt = torch.empty(2 * c, c, 2, 2, device=DEV)
    _lib.check(lib.ammc_unpack_convt_wgrad_f32(_ptr(dwp), 2 * c, c, _ptr(dwt), s), "unpack_convt")
    wt = torch.zeros(2 * c, c, 2, 2, dtype=torch.float64, requires_grad=True)
    (F.conv_transpose2d(x.double().cpu().permute(0, 3, 1, 2), wt, stride=2) * dy.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    err = float((dwt.double().cpu() - wt.grad).abs().max() / wt.grad.abs().max())
    assert err <= 3e-6, err


@pytest.mark.parametrize("y_f32", [0, 1])
def test_conv_gemm_s16_4x4_stride2_lrelu(y_f32):
    """Conv2d(k 4, stride 2, padding 2) + bias + LeakyReLU(0.1) (pix2pix_networks.py:604-606) on the split-fp16 kernel"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    B, H, W, cin, n = 2, 21, 30, 64, 128
    oh, ow = (H + 4 - 4) // 2 + 1, (W + 4 - 4) // 2 + 1
    x = S.hashed_uniform("c4x", (B, H, W, cin)).to(DEV)
    w = (S.hashed_uniform("c4w", (n, cin, 4, 4)) * 0.05).to(DEV)
    bias = S.hashed_uniform("c4b", (n,)).to(DEV)
    X32 = Act(torch.zeros(B, H + 4, W + 4, cin, device=DEV), B, H, W, cin, 0, 2)
    X32.interior().copy_(x)
    X16 = _s16(lib, X32.buf)
    wp = torch.zeros(n, _kpad(16 * cin), device=DEV)
    _lib.check(lib.ammc_pack_conv_weight_f32(_ptr(w), n, cin, 4, cin, _ptr(wp), s), "pack")
    w16 = _s16(lib, wp)
    Y = Act(torch.zeros(B, oh + 2, ow + 2, n, device=DEV), B, oh, ow, n, 0, 1)
    d = AmmcConvDesc()
    d.x, d.w, d.y, d.shift = _ptr(X16), _ptr(w16), Y.pix0(), _ptr(bias)
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act, d.x_step, d.y_f32 = B, oh, ow, cin, 16, n, 1, n, ACT_LRELU, 2, y_f32
    d.x_bs, d.x_rs, d.x_ps = X32.strides
    d.y_bs, d.y_rs, d.y_ps = Y.strides
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv_gemm_s16(4x4)")
    want = F.leaky_relu(F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), bias.double().cpu(), stride=2, padding=2), 0.1)
    got = (Y.interior().double() if y_f32 else _decode(Y.buf, n)[:, 1:-1, 1:-1]).cpu().permute(0, 3, 1, 2)
    assert float((got - want).abs().max() / want.abs().max()) <= 3e-6


def test_pack_filters_s16_equals_pack_then_split():
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    shapes = [(64, 12, 16), (64, 64, 64), (128, 64, 64), (256, 128, 128), (512, 512, 512), (64, 128, 128)]     # (cout, cin, cin_p)
    dt = np.dtype([("w", "u8"), ("out", "u8"), ("cout", "i4"), ("cin", "i4"), ("inner_p", "i4"), ("kpad", "i4"),
                   ("kind", "i4"), ("rows", "i4"), ("group_end", "i8")])
    assert dt.itemsize == lib.ammc_pack_filters_item_bytes()
    items, want, keep, total = [], [], [], 0
    for i, (cout, cin, cin_p) in enumerate(shapes):
        w = S.hashed_normal(f"pk{i}", (cout, cin, 3, 3), 0.05).to(DEV)
        kpad = _kpad(9 * cin_p)
        wp = torch.zeros(cout, kpad, device=DEV)
        _lib.check(lib.ammc_pack_conv_weight_f32(_ptr(w), cout, cin, 3, cin_p, _ptr(wp), s), "pack")
        out = torch.full((cout, kpad), 7.0, device=DEV)
        total += cout * kpad // 8
        items.append((w.data_ptr(), out.data_ptr(), cout, cin, cin_p, kpad, 0, cout, total))
        want.append(_s16(lib, wp))
        keep.append((w, out))
        if cin >= 32:                                  # the flipped / transposed input-gradient filter of the same layer
            rows, kd = max(64, (cin + 63) // 64 * 64), _kpad(9 * cout)
            wd = torch.zeros(rows, kd, device=DEV)
            _lib.check(lib.ammc_pack_conv_dgrad_weight_f32(_ptr(w), cout, cin, cout, rows, _ptr(wd), s), "pack_dgrad")
            out = torch.full((rows, kd), 7.0, device=DEV)
            total += rows * kd // 8
            items.append((w.data_ptr(), out.data_ptr(), cout, cin, cout, kd, 1, rows, total))
            want.append(_s16(lib, wd))
            keep.append((w, out))
    table = torch.from_numpy(np.array(items, dtype=dt).view(np.uint8).copy()).to(DEV)
    _lib.check(lib.ammc_pack_filters_s16(table.data_ptr(), len(items), total, s), "pack_filters_s16")
    for (w, out), ref in zip(keep, want):
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))


def test_lrelu_s16_in_place_on_a_channel_slice():
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    B, H, W, ctot, c0, c = 2, 5, 7, 96, 32, 40
    x = (S.hashed_normal("lr16", (B, H + 2, W + 2, ctot), 3.0)).to(DEV)
    x16 = _s16(lib, x)
    a = Act(x16, B, H, W, c, c0, 1)
    _lib.check(lib.ammc_lrelu_s16(a.pix0(), *a.strides, B, H, W, c, 0.1, s), "lrelu_s16")
    got = _decode(x16, ctot)
    ref = _decode(_s16(lib, x), ctot)
    want = ref.clone()
    sl = want[:, 1:-1, 1:-1, c0:c0 + c]
    want[:, 1:-1, 1:-1, c0:c0 + c] = torch.where(sl > 0, sl, 0.1 * sl)
    assert float((got - want).abs().max()) <= 3e-7 * float(want.abs().max())
    mask = torch.ones_like(want, dtype=torch.bool)
    mask[:, 1:-1, 1:-1, c0:c0 + c] = False
    assert torch.equal(got[mask], ref[mask])                      # everything outside the slice's interior is untouched


@pytest.mark.parametrize("B,H,W,cin,n,want", [(3, 128, 128, 64, 64, "conv_tap_s16<4, 1, 2, 2, 1, 0, 1>+stats"),
                                              (4, 256, 256, 32, 128, "conv_tap_s16<4, 1, 2, 4, 1, 0, 1>+stats"),
                                              (2, 64, 64, 64, 256, None)])
def test_conv_s16_statistics_output_is_the_channel_sums_of_what_it_stored(B, H, W, cin, n, want):
    """AmmcConvDesc.stats (training-mode BatchNorm statistics as a second output of the convolution, csrc/conv_tap_s16.hip):
    one row per 8 x 32 output patch = that patch's per-channel sum and sum of squares of the fp32 values the kernel
    stored, and `ammc_bn_finalize_f32` over the rows = mean / biased variance of the tensor.  A kernel without the
    epilogue reports 0 rows and refuses a descriptor that asks for it."""
    from ammcnet_aaai2021_amd.engine import s16_variant
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tag = f"cstat-{B}-{H}-{W}-{cin}-{n}"
    X = Act(torch.zeros(B, H + 2, W + 2, cin, device=DEV), B, H, W, cin, 0, 1)
    X.interior().copy_(S.hashed_uniform(tag + "x", (B, H, W, cin)).to(DEV) + 0.25)       # non-zero mean
    w = (S.hashed_uniform(tag + "w", (n, 9 * cin)) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    X16, w16 = Act(_s16(lib, X.buf), B, H, W, cin, 0, 1), _s16(lib, w)
    Y = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
    d = AmmcConvDesc()
    d.x, d.w, d.y = X16.tap0(), _ptr(w16), Y.pix0()
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act, d.y_f32, d.x_step = B, H, W, cin, 9, n, 1, n, 0, 1, 1
    d.x_bs, d.x_rs, d.x_ps = X16.strides
    d.y_bs, d.y_rs, d.y_ps = Y.strides
    rows = lib.ammc_conv_gemm_s16_stats_rows(C.byref(d))
    if want is None:
        assert rows == 0
        d.stats = _ptr(torch.zeros(8, device=DEV))
        assert lib.ammc_conv_gemm_s16(C.byref(d), s) == -2                      # AMMC_EUNSUP, nothing launched
        return
    assert rows == B * (H // 8) * (W // 32)
    stats = torch.full((rows, 2, n), float("nan"), device=DEV)
    d.stats = _ptr(stats)
    assert s16_variant(d) == want
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv+stats")
    torch.cuda.synchronize()
    y = Y.interior().double()                                                        # what the kernel stored
    patches = y.reshape(B, H // 8, 8, W // 32, 32, n).permute(0, 1, 3, 2, 4, 5).reshape(rows, 256, n)
    ref1, ref2 = patches.sum(1), (patches * patches).sum(1)
    got = stats.double()
    assert bool(torch.isfinite(got).all())
    e1 = float((got[:, 0] - ref1).abs().max() / ref1.abs().max())
    e2 = float((got[:, 1] - ref2).abs().max() / ref2.abs().max())
    assert e1 <= 2e-6 and e2 <= 2e-6, (e1, e2)                                      # fp32 sums of 256 values
    # the same output without statistics is bit-identical (the epilogue only adds a second output)
    Y2 = Act(torch.zeros_like(Y.buf), B, H, W, n, 0, 1)
    d.y, d.stats = Y2.pix0(), None
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv")
    assert torch.equal(Y2.buf, Y.buf)
    # ... and the finalizer over the rows gives the tensor's batch statistics
    gamma, beta = torch.ones(n, device=DEV), torch.zeros(n, device=DEV)
    rm, rv = torch.zeros(n, device=DEV), torch.ones(n, device=DEV)
    mean, invstd, scale, shift = (torch.empty(n, device=DEV) for _ in range(4))
    _lib.check(lib.ammc_bn_finalize_f32(_ptr(stats), rows, n, float(B * H * W), _ptr(gamma), _ptr(beta), 1e-5, 0.1, _ptr(rm),
                                        _ptr(rv), _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), s), "bn_finalize")
    flat = y.reshape(-1, n)
    assert float((mean.double() - flat.mean(0)).abs().max()) <= 1e-6 * float(flat.abs().max())
    ref_is = 1.0 / torch.sqrt(flat.var(0, unbiased=False) + 1e-5)
    assert float(((invstd.double() - ref_is) / ref_is).abs().max()) <= 1e-5


def test_reduce_partials_in_segments():
    """`ammc_reduce_partials_seg_f32`: out[s] = the sum of rows [s * seg, (s + 1) * seg) (the last run may be short)"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    rows, qc, seg = 1000, 136, 128
    part = S.hashed_uniform("rpseg", (rows, qc)).to(DEV)
    nseg = (rows + seg - 1) // seg
    out = torch.full((nseg, qc), float("nan"), device=DEV)
    _lib.check(lib.ammc_reduce_partials_seg_f32(_ptr(part), rows, qc, seg, qc, _ptr(out), s), "reduce_seg")
    ref = torch.stack([part[i * seg:(i + 1) * seg].double().sum(0) for i in range(nseg)])
    assert float((out.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    # columns from max_from on: the maximum of the run (exact)
    _lib.check(lib.ammc_reduce_partials_seg_f32(_ptr(part), rows, qc, seg, 64, _ptr(out), s), "reduce_seg(max)")
    refm = torch.stack([part[i * seg:(i + 1) * seg].max(0).values for i in range(nseg)])
    assert float((out[:, :64].double() - ref[:, :64]).abs().max()) <= 1e-6 * float(ref.abs().max())
    assert torch.equal(out[:, 64:], refm[:, 64:])


@pytest.mark.parametrize("B,fh,fw,c", [(2, 16, 24, 64), (3, 9, 13, 16), (1, 64, 64, 128)])
def test_maxpool_index_map_routes_like_the_recomputed_maxima(B, fh, fw, c):
    """`ammc_maxpool2x2_s16_idx` = `ammc_maxpool2x2_s16` + a byte per pooled element (the window position of the first
    maximum); `ammc_maxpool2x2_bwd_idx_f32` from those bytes = `ammc_maxpool2x2_bwd_s16x_f32`, which finds the maxima
    again from the pooled tensor - bit for bit, odd sizes (a last row / column outside every window) included"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    h, w = fh // 2, fw // 2
    tag = f"mpidx-{B}-{fh}-{fw}-{c}"
    x = S.hashed_uniform(tag + "x", (B, fh, fw, c))
    x[:, 0:2, 0:2, :8] = 0.5                                    # a tie in the first window: the first position wins
    X = Act(torch.zeros(B, fh + 2, fw + 2, c, device=DEV), B, fh, fw, c, 0, 1)
    X.interior().copy_(x.to(DEV))
    X16 = Act(_s16(lib, X.buf), B, fh, fw, c, 0, 1)
    P_a = Act(torch.zeros(B, h + 2, w + 2, c, device=DEV), B, h, w, c, 0, 1)
    P_b = Act(torch.zeros_like(P_a.buf), B, h, w, c, 0, 1)
    idx = torch.full((B, h, w, c), 255, dtype=torch.uint8, device=DEV)
    _lib.check(lib.ammc_maxpool2x2_s16(X16.pix0(), *X16.strides, P_a.pix0(), *P_a.strides, B, h, w, c, s), "pool")
    _lib.check(lib.ammc_maxpool2x2_s16_idx(X16.pix0(), *X16.strides, P_b.pix0(), *P_b.strides, idx.data_ptr(), B, h, w, c, s), "pool+idx")
    assert torch.equal(P_a.buf, P_b.buf)
    xd = _decode(X16.buf, c)[:, 1:-1, 1:-1][:, :2 * h, :2 * w]
    win = xd.reshape(B, h, 2, w, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(B, h, w, 4, c)
    assert torch.equal(idx.long(), win.argmax(3)) or torch.equal(win.gather(3, idx.long()[:, :, :, None]).squeeze(3), win.max(3).values)
    assert int(idx[:, 0, 0, :8].max()) == 0
    dp = Act(torch.zeros(B, h + 2, w + 2, c, device=DEV), B, h, w, c, 0, 1)
    dp.interior().copy_(S.hashed_uniform(tag + "g", (B, h, w, c)).to(DEV))
    add = Act(torch.zeros(B, fh + 2, fw + 2, c, device=DEV), B, fh, fw, c, 0, 1)
    add.interior().copy_(S.hashed_uniform(tag + "a", (B, fh, fw, c)).to(DEV))
    outs = []
    for which in range(2):
        out = Act(torch.full((B, fh + 2, fw + 2, c), 7.0, device=DEV), B, fh, fw, c, 0, 1)
        if which == 0:
            _lib.check(lib.ammc_maxpool2x2_bwd_s16x_f32(X16.pix0(), *X16.strides, dp.pix0(), *dp.strides, add.pix0(), *add.strides,
                                                        out.pix0(), *out.strides, B, h, w, fh, fw, c, s), "bwd s16x")
        else:
            _lib.check(lib.ammc_maxpool2x2_bwd_idx_f32(idx.data_ptr(), dp.pix0(), *dp.strides, add.pix0(), *add.strides,
                                                       out.pix0(), *out.strides, B, h, w, fh, fw, c, s), "bwd idx")
        outs.append(out.buf.clone())
    assert torch.equal(outs[0], outs[1])
    # ... and it is MaxPool2d's gradient plus `add`
    xt = xd.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    F.max_pool2d(xt, 2).backward(dp.interior().double().permute(0, 3, 1, 2))
    ref = add.interior().double().clone()
    ref[:, :2 * h, :2 * w] += xt.grad.permute(0, 2, 3, 1)
    got = outs[1][:, 1:-1, 1:-1].double()
    mism = (got - ref).abs() > 1e-6
    mism[:, 0:2, 0:2, :8] = False                               # (the planted tie: whichever position torch picks)
    assert not bool(mism.any()), int(mism.sum())


@pytest.mark.parametrize("B,H,W,c,cbuf,relu", [(2, 16, 32, 64, 128, 1), (3, 8, 12, 128, 128, 1), (1, 64, 64, 16, 16, 0)])
def test_bn_apply_with_pooled_output_equals_apply_then_pool(B, H, W, c, cbuf, relu):
    """`ammc_scale_shift_act_s16_pool_f32` (the apply pass of a unit that a MaxPool2d(2) follows) = `ammc_scale_shift_act_s16_f32`
    + `ammc_maxpool2x2_s16_idx` on its S16 output, bit for bit: y16 (written into a channel slice of a wider buffer, as the
    skip tensors are), the fp32 y when asked for, the pooled S16 tensor and the window positions"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tag = f"applypool-{B}-{H}-{W}-{c}"
    X = Act(torch.zeros(B, H + 2, W + 2, c, device=DEV), B, H, W, c, 0, 1)
    X.interior().copy_((S.hashed_uniform(tag + "x", (B, H, W, c)) - 0.5).to(DEV))
    scale = (S.hashed_uniform(tag + "s", (c,)) + 0.5).to(DEV)
    shift = (S.hashed_uniform(tag + "h", (c,)) - 0.5).to(DEV)
    h, w = H // 2, W // 2
    res = []
    for fused in (False, True):
        Y32 = Act(torch.zeros(B, H + 2, W + 2, cbuf, device=DEV), B, H, W, c, 0, 1)
        Y16 = Act(torch.zeros_like(Y32.buf), B, H, W, c, 0, 1)
        P16 = Act(torch.zeros(B, h + 2, w + 2, c, device=DEV), B, h, w, c, 0, 1)
        idx = torch.full((B, h, w, c), 255, dtype=torch.uint8, device=DEV)
        if fused:
            _lib.check(lib.ammc_scale_shift_act_s16_pool_f32(X.pix0(), *X.strides, _ptr(scale), _ptr(shift), Y32.pix0(), Y16.pix0(),
                                                             *Y16.strides, P16.pix0(), *P16.strides, idx.data_ptr(), relu, B, H, W, c, s),
                       "apply+pool")
        else:
            _lib.check(lib.ammc_scale_shift_act_s16_f32(X.pix0(), *X.strides, _ptr(scale), _ptr(shift), None, 0, 0, 0, Y32.pix0(),
                                                        Y16.pix0(), *Y16.strides, relu, B, H, W, c, s), "apply")
            _lib.check(lib.ammc_maxpool2x2_s16_idx(Y16.pix0(), *Y16.strides, P16.pix0(), *P16.strides, idx.data_ptr(), B, h, w, c, s), "pool")
        res.append((Y32.buf.clone(), Y16.buf.clone(), P16.buf.clone(), idx.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[1][0].abs().max()) > 0 and int(res[1][3].max()) <= 3
    # an odd size is refused (MaxPool2d floors: the two passes handle it)
    assert lib.ammc_scale_shift_act_s16_pool_f32(X.pix0(), *X.strides, _ptr(scale), _ptr(shift), None, Y16.pix0(), *Y16.strides,
                                                 P16.pix0(), *P16.strides, idx.data_ptr(), relu, B, H - 1, W, c, s) == -2


@pytest.mark.parametrize("B,H,W,c,cbuf", [(2, 16, 32, 64, 128), (3, 9, 13, 16, 16), (1, 32, 32, 128, 128)])
def test_bn_backward_with_the_max_pool_gradient_formed_on_the_fly(B, H, W, c, cbuf):
    """`ammc_bn_bwd_reduce_bound_unpool_f32` / `ammc_bn_bwd_apply_s16_unpool_f32` (dy = add + MaxPool2d-backward(dpo) by the
    recorded window positions, never written) = `ammc_maxpool2x2_bwd_idx_f32` into a buffer, then the plain passes on it:
    partial rows, sums and the S16 gradient bit for bit - `add` in a channel slice of a wider buffer (the concat gradient),
    an odd size (a last row / column outside every window) included"""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tag = f"unpoolbn-{B}-{H}-{W}-{c}"
    ph, pw = H // 2, W // 2
    craw = Act(torch.zeros(B, H + 2, W + 2, c, device=DEV), B, H, W, c, 0, 1)
    craw.interior().copy_((S.hashed_uniform(tag + "c", (B, H, W, c)) - 0.5).to(DEV))
    add = Act(torch.zeros(B, H + 2, W + 2, cbuf, device=DEV), B, H, W, c, 0, 1)
    add.interior().copy_(((S.hashed_uniform(tag + "a", (B, H, W, c)) - 0.5) * 1e-3).to(DEV))
    dpo = Act(torch.zeros(B, ph + 2, pw + 2, c, device=DEV), B, ph, pw, c, 0, 1)
    dpo.interior().copy_(((S.hashed_uniform(tag + "p", (B, ph, pw, c)) - 0.5) * 1e-3).to(DEV))
    idx = (S.hashed_uniform(tag + "i", (B, ph, pw, c)) * 4).clamp(0, 3).to(torch.uint8).to(DEV).contiguous()
    mean = (S.hashed_uniform(tag + "m", (c,)) - 0.5).to(DEV)
    invstd = (S.hashed_uniform(tag + "v", (c,)) + 1.0).to(DEV)
    scale = ((S.hashed_uniform(tag + "g", (c,)) + 0.5) * invstd.cpu()).to(DEV)
    shift = (S.hashed_uniform(tag + "b", (c,)) - 0.5).to(DEV)
    assert lib.ammc_bn_bwd_unpool_supported(c, craw.ps, add.ps, c, W) == 1
    nblk = lib.ammc_chan_reduce_blocks(B * H * W)
    outs = []
    for fused in (False, True):
        partial = torch.full((nblk, 4, c), float("nan"), device=DEV)
        sums = torch.empty(2 * c, device=DEV)
        amax = torch.zeros(256, dtype=torch.int32, device=DEV)
        inv = torch.empty(8, device=DEV)
        dc16 = Act(torch.zeros(B, H + 2, W + 2, c, device=DEV), B, H, W, c, 0, 1)
        stats = (_ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift))
        up = (dpo.pix0(), *dpo.strides, idx.data_ptr(), ph, pw)
        if fused:
            _lib.check(lib.ammc_bn_bwd_reduce_bound_unpool_f32(craw.pix0(), *craw.strides, add.pix0(), *add.strides, *up, *stats, 1,
                                                               B, H, W, c, _ptr(partial), s), "reduce(unpool)")
        else:
            dy = Act(torch.zeros(B, H + 2, W + 2, c, device=DEV), B, H, W, c, 0, 1)
            _lib.check(lib.ammc_maxpool2x2_bwd_idx_f32(idx.data_ptr(), dpo.pix0(), *dpo.strides, add.pix0(), *add.strides,
                                                       dy.pix0(), *dy.strides, B, ph, pw, H, W, c, s), "maxpool_bwd")
            _lib.check(lib.ammc_bn_bwd_reduce_bound_f32(craw.pix0(), *craw.strides, dy.pix0(), *dy.strides, *stats, 1, B, H, W, c,
                                                        _ptr(partial), s), "reduce")
        _lib.check(lib.ammc_bn_bwd_finalize_f32(_ptr(partial), nblk, c, B * H * W, _ptr(scale), _ptr(sums), amax.data_ptr(), s), "finalize")
        if fused:
            _lib.check(lib.ammc_bn_bwd_apply_s16_unpool_f32(craw.pix0(), *craw.strides, add.pix0(), *add.strides, *up, *stats,
                                                            _ptr(sums), 1, dc16.pix0(), None, *dc16.strides, B, H, W, c,
                                                            amax.data_ptr(), _ptr(inv), 8, s), "apply(unpool)")
        else:
            _lib.check(lib.ammc_bn_bwd_apply_s16_f32(craw.pix0(), *craw.strides, dy.pix0(), *dy.strides, *stats, _ptr(sums), 1,
                                                     dc16.pix0(), None, *dc16.strides, B, H, W, c, amax.data_ptr(), _ptr(inv), 8, s),
                       "apply")
        outs.append((partial.clone(), sums.clone(), dc16.buf.clone(), inv.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(_decode(outs[1][2], c).abs().max()) > 0


@pytest.mark.parametrize("B,H,W,cin,n", [(3, 128, 128, 64, 64), (4, 256, 256, 32, 128)])
def test_conv_s16_batchnorm_backward_statistics_output(B, H, W, cin, n):
    """AmmcConvDesc.bn_c: the input-gradient convolution of a training backward also leaves the BatchNorm-backward partial
    rows of the unit its output goes to - per patch: sum g, sum g xhat, max |g|, max |xhat| with g = y [c scale + shift > 0],
    xhat = (c - mean) invstd, c = that unit's saved convolution output.  Combined over the patches they equal what
    `ammc_bn_bwd_reduce_bound_f32` finds by reading y back (sums to fp32 rounding of a different partition, maxima
    exactly), and y itself is bit-identical to the plain launch."""
    from ammcnet_aaai2021_amd.engine import s16_variant
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tag = f"cbnb-{B}-{H}-{W}-{cin}-{n}"
    X = Act(torch.zeros(B, H + 2, W + 2, cin, device=DEV), B, H, W, cin, 0, 1)
    X.interior().copy_((S.hashed_uniform(tag + "x", (B, H, W, cin)) - 0.5).to(DEV))
    w = ((S.hashed_uniform(tag + "w", (n, 9 * cin)) - 0.5) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    X16, w16 = Act(_s16(lib, X.buf), B, H, W, cin, 0, 1), _s16(lib, w)
    craw = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
    craw.interior().copy_((S.hashed_uniform(tag + "c", (B, H, W, n)) - 0.5).to(DEV))
    mean = (S.hashed_uniform(tag + "m", (n,)) - 0.5).to(DEV)
    invstd = (S.hashed_uniform(tag + "v", (n,)) + 1.0).to(DEV)
    scale = ((S.hashed_uniform(tag + "g", (n,)) + 0.5) * invstd.cpu()).to(DEV)
    shift = (S.hashed_uniform(tag + "b", (n,)) - 0.5).to(DEV)
    inv = torch.full((n,), 0.125, device=DEV)                     # the epilogue scale of a rescaled gradient operand
    Y = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
    d = AmmcConvDesc()
    d.x, d.w, d.y, d.scale = X16.tap0(), _ptr(w16), Y.pix0(), _ptr(inv)
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act, d.y_f32, d.x_step = B, H, W, cin, 9, n, 1, n, 0, 1, 1
    d.x_bs, d.x_rs, d.x_ps = X16.strides
    d.y_bs, d.y_rs, d.y_ps = Y.strides
    d.bn_c = craw.pix0()
    d.bn_bs, d.bn_rs, d.bn_ps = craw.strides
    d.bn_mean, d.bn_invstd, d.bn_scale, d.bn_shift, d.bn_relu = _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), 1
    rows = lib.ammc_conv_gemm_s16_stats_rows(C.byref(d))
    assert rows == B * (H // 8) * (W // 32)
    stats = torch.full((rows, 4, n), float("nan"), device=DEV)
    d.stats = _ptr(stats)
    assert s16_variant(d).endswith("+bnbwd")
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "dgrad+bnbwd")
    nblk = lib.ammc_chan_reduce_blocks(B * H * W)
    partial = torch.empty(nblk, 4, n, device=DEV)
    _lib.check(lib.ammc_bn_bwd_reduce_bound_f32(craw.pix0(), *craw.strides, Y.pix0(), *Y.strides, _ptr(mean), _ptr(invstd), _ptr(scale),
                                                _ptr(shift), 1, B, H, W, n, _ptr(partial), s), "reduce")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(stats).all())
    got_s, ref_s = stats[:, :2].double().sum(0), partial[:, :2].double().sum(0)
    assert float((got_s - ref_s).abs().max()) <= 2e-6 * float(ref_s.abs().max()) + 1e-9, float((got_s - ref_s).abs().max())
    assert torch.equal(stats[:, 2:].max(0).values, partial[:, 2:].max(0).values)
    # per patch against fp64 of the stored y
    y, c = Y.interior().double(), craw.interior().double()
    g = torch.where(c * scale.double() + shift.double() > 0, y, torch.zeros_like(y))
    xh = (c - mean.double()) * invstd.double()
    def patches(t):
        return t.reshape(B, H // 8, 8, W // 32, 32, n).permute(0, 1, 3, 2, 4, 5).reshape(rows, 256, n)
    refp = torch.stack([patches(g).sum(1), patches(g * xh).sum(1), patches(g.abs()).max(1).values, patches(xh.abs()).max(1).values], 1)
    err = (stats.double() - refp).abs().amax((0, 2)) / refp.abs().amax((0, 2))
    assert float(err.max()) <= 3e-6, err
    # the two-stage combination keeps sums and maxima apart
    seg = torch.empty((rows + 127) // 128, 4, n, device=DEV)
    _lib.check(lib.ammc_reduce_partials_seg_f32(_ptr(stats), rows, 4 * n, 128, 2 * n, _ptr(seg), s), "reduce_seg")
    assert torch.equal(seg[:, 2:].max(0).values, stats[:, 2:].max(0).values)
    assert float((seg[:, :2].double().sum(0) - got_s).abs().max()) <= 1e-6 * float(got_s.abs().max()) + 1e-9
    # same y without the statistics
    Y2 = Act(torch.zeros_like(Y.buf), B, H, W, n, 0, 1)
    d.y, d.stats, d.bn_c = Y2.pix0(), None, None
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "dgrad")
    assert torch.equal(Y2.buf, Y.buf)
