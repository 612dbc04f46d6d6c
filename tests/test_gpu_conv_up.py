"""The fused decoder step (csrc/conv_up_s16.hip): ConvTranspose2d(k 2, s 2) + cat([skip, .]) + conv3x3 + BN(eval) + ReLU
as ONE launch, the transposed conv folded into the 3x3 taps per output-pixel parity (`ammc_pack_up_conv_f32`) - against
an fp64 evaluation of the reference's operator sequence (`up.forward`, models/unet.py:50-59, + the first conv of
`up.conv`, :11-13) on the operands as the kernel sees them (S16-rounded activations).  Small images, so every border
class of the composed bias is exercised.  3e-6 of max|ref|."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from ammcnet_aaai2021_amd import _lib, synthetic as S
from ammcnet_aaai2021_amd._lib import ACT_NONE, ACT_RELU, AmmcConvDesc
from ammcnet_aaai2021_amd.engine import Act, _ptr
from test_gpu_conv_tap import _s16_act, _s16_read

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,H,W,c,n,relu", [
    (2, 16, 32, 64, 64, True),          # up3 of the network: 64 filters
    (1, 8, 64, 128, 128, True),         # up2: 128 filters (two 64-filter halves in phase B)
    (2, 16, 32, 256, 256, False),       # up1: two N tiles
    (3, 24, 96, 32, 64, True),          # one skip channel block, several patches per image
])
def test_conv_up_s16_vs_fp64(B, H, W, c, n, relu):
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tag = f"up-{B}-{H}-{W}-{c}-{n}"
    x2 = S.hashed_uniform(tag + "x2", (B, 2 * c, H // 2, W // 2)).to(DEV)
    skip = S.hashed_uniform(tag + "sk", (B, c, H, W)).to(DEV)
    w3 = (S.hashed_uniform(tag + "w3", (n, 2 * c, 3, 3)) * (2.0 / (18 * c)) ** 0.5).to(DEV)
    wt = (S.hashed_uniform(tag + "wt", (2 * c, c, 2, 2)) * (1.0 / (2 * c)) ** 0.5).to(DEV)
    bt = (S.hashed_uniform(tag + "bt", (c,)) * 0.3).to(DEV)
    scale = S.hashed_uniform(tag + "s", (n,), 0.7, 1.3).to(DEV)
    shift = S.hashed_uniform(tag + "b", (n,), -0.2, 0.2).to(DEV)
    x2a = _s16_act(x2)
    ska = _s16_act(skip, 2 * c, 0)                      # the skip tensor is the first half of the concat buffer
    ya = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
    # the whole 3x3 filter, packed and split as for ammc_conv_gemm_s16
    wp = torch.empty(n, 9 * 2 * c, device=DEV)
    _lib.check(lib.ammc_pack_conv_weight_f32(_ptr(w3), n, 2 * c, 3, 2 * c, _ptr(wp), s), "pack")
    ws = torch.empty_like(wp)
    _lib.check(lib.ammc_split_rows_f32(_ptr(wp), wp.numel(), _ptr(ws), s), "split")
    w2 = torch.empty(n, 16 * 2 * c, device=DEV)
    shift9 = torch.empty(9, n, device=DEV)
    _lib.check(lib.ammc_pack_up_conv_f32(_ptr(w3), _ptr(wt), _ptr(bt), _ptr(scale), _ptr(shift), n, c, _ptr(w2),
                                         _ptr(shift9), s), "pack_up")
    w2s = torch.empty_like(w2)
    _lib.check(lib.ammc_split_rows_f32(_ptr(w2), w2.numel(), _ptr(w2s), s), "split")
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    d = AmmcConvDesc()
    d.x, d.w, d.y, d.scale = ska.tap0(), _ptr(ws), ya.pix0(), _ptr(scale)
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup = B, H, W, c, 9, n, 1, n
    d.act = ACT_RELU if relu else ACT_NONE
    d.x_bs, d.x_rs, d.x_ps = ska.strides
    d.y_bs, d.y_rs, d.y_ps = ya.strides
    d.overflow_flag = flag.data_ptr()
    _lib.check(lib.ammc_conv_up_s16(C.byref(d), x2a.tap0(), *x2a.strides, 2 * c, _ptr(w2s), _ptr(shift9), s), "conv_up_s16")
    got = _s16_read(ya).double().cpu()
    # the reference's operator sequence in fp64
    x2r, skr = _s16_read(x2a).double().cpu(), _s16_read(ska).double().cpu()
    up = F.conv_transpose2d(x2r, wt.double().cpu(), bt.double().cpu(), stride=2)
    want = F.conv2d(torch.cat([skr, up], 1), w3.double().cpu(), padding=1)
    want = want * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)
    if relu:
        want = want.clamp_min(0)
    err = float((got - want).abs().max() / want.abs().max())
    assert err <= 3e-6, err
    assert int(flag.item()) == 0
    assert float(ya.buf[:, 0].abs().max()) == 0.0 and float(ya.buf[:, :, 0].abs().max()) == 0.0     # halo untouched


def test_conv_up_s16_rejects_what_it_does_not_take():
    lib = _lib.load()
    d = AmmcConvDesc()
    d.x, d.w, d.y = 0x100000, 0x200000, 0x300000
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup = 1, 8, 24, 64, 9, 64, 1, 64       # W % 32 != 0
    d.x_ps, d.x_rs, d.x_bs = 128, 26 * 128, 10 * 26 * 128
    d.y_ps, d.y_rs, d.y_bs = 64, 26 * 64, 10 * 26 * 64
    assert lib.ammc_conv_up_s16(C.byref(d), 0x400000, 8, 8, 8, 128, 0x500000, 0x600000, None) == -2
    assert lib.ammc_conv_up_s16(C.byref(d), None, 8, 8, 8, 128, 0x500000, 0x600000, None) == -1


def _s16_round(t: torch.Tensor) -> torch.Tensor:
    """the value an S16 pair (hi, lo) carries for an fp32 number (what the kernels' operands are), in float64"""
    hi = t.half().float()
    lo = ((t - hi) * 2048.0).half().float()
    return hi.double() + lo.double() / 2048.0


@pytest.mark.parametrize("B,C,H,W", [(2, 12, 16, 32), (1, 6, 8, 64), (3, 12, 24, 96), (2, 3, 8, 32), (5, 12, 256, 256), (9, 6, 136, 256)])
def test_conv_first_s16_vs_fp64(B, C, H, W):
    """csrc/conv_first_s16.hip: `inconv`'s first conv + BN(eval) + ReLU straight from the NCHW fp32 clips (zero padding,
    S16 split and im2col inside the kernel; reference unet.py:11-13, 23-30) against an fp64 convolution of the
    S16-rounded operands; small images, so every patch touches the image border, and two sizes with more tiles than
    the persistent grid of 512 workgroups (1280 and 1224: workgroups with two and with three tiles)."""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tag = f"first-{B}-{C}-{H}-{W}"
    x = (S.hashed_uniform(tag + "x", (B, C, H, W)) * 1.7).to(DEV)
    w = (S.hashed_uniform(tag + "w", (64, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).to(DEV)
    scale = S.hashed_uniform(tag + "s", (64,), 0.7, 1.3).to(DEV)
    shift = S.hashed_uniform(tag + "b", (64,), -0.2, 0.2).to(DEV)
    img = torch.empty(lib.ammc_first_conv_image_floats(), device=DEV)
    _lib.check(lib.ammc_pack_first_conv_f32(_ptr(w), 64, C, _ptr(img), s), "pack_first")
    ya = Act(torch.zeros(B, H + 2, W + 2, 64, device=DEV), B, H, W, 64, 0, 1)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(lib.ammc_conv_first_s16(_ptr(x), B, C, H, W, _ptr(img), _ptr(scale), _ptr(shift), ACT_RELU, ya.pix0(),
                                       *ya.strides, flag.data_ptr(), s), "conv_first_s16")
    got = _s16_read(ya).double().cpu()
    want = F.conv2d(_s16_round(x.cpu()), _s16_round(w.cpu()), padding=1)
    want = (want * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)).clamp_min(0)
    err = float((got - want).abs().max() / want.abs().max())
    assert err <= 3e-6, err
    assert int(flag.item()) == 0
    assert float(ya.buf[:, 0].abs().max()) == 0.0 and float(ya.buf[:, :, 0].abs().max()) == 0.0     # halo untouched
    assert lib.ammc_conv_first_s16(_ptr(x), B, 17, H, W, _ptr(img), None, None, ACT_RELU, ya.pix0(), *ya.strides, None, s) == -2
