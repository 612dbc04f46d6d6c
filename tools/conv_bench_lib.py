"""descriptor of one isolated S16 3x3 layer on random operands (shared by tools/conv_bench.py and tools/micro/*)"""
import torch
from ammcnet_aaai2021_amd import _lib
from ammcnet_aaai2021_amd._lib import ACT_RELU, AmmcConvDesc
from ammcnet_aaai2021_amd.engine import Act, _ptr

dev = "cuda:0"


def s16_of(t):
    lib = _lib.load()
    out = torch.empty_like(t)
    _lib.check(lib.ammc_split_rows_f32(_ptr(t), t.numel(), _ptr(out), torch.cuda.current_stream().cuda_stream), "split")
    return out


def make_desc(B, H, W, cin, n, const=False):
    """const: every activation 0.5, every filter tap 0.01 (no operand toggling: the chip holds a higher clock)"""
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x32 = torch.zeros(B, H + 2, W + 2, cin, device=dev)
    if const:
        x32[:, 1:-1, 1:-1] = 0.5
    else:
        x32[:, 1:-1, 1:-1].copy_(torch.randn(B, H, W, cin, device=dev, generator=g))
    xa = Act(s16_of(x32), B, H, W, cin, 0, 1)
    del x32
    ya = Act(torch.zeros(B, H + 2, W + 2, n, device=dev), B, H, W, n, 0, 1)
    w32 = torch.full((n, 9 * cin), 0.01, device=dev) if const else torch.randn(n, 9 * cin, device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5
    ws = s16_of(w32)
    scale = torch.ones(n, device=dev)
    shift = torch.zeros(n, device=dev)
    d = AmmcConvDesc()
    d.x, d.w, d.y, d.scale, d.shift = xa.tap0(), _ptr(ws), ya.pix0(), _ptr(scale), _ptr(shift)
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act = B, H, W, cin, 9, n, 1, n, ACT_RELU
    d.x_bs, d.x_rs, d.x_ps = xa.strides
    d.y_bs, d.y_rs, d.y_ps = ya.strides
    return d, (xa, ya, ws, scale, shift)
