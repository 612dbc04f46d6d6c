"""Would the cross terms of the split-fp16 product survive fp8?  (the one arithmetic lever left.)

The S16 kernels compute  x w = x_hi w_hi + (x_hi w_lo + x_lo w_hi)  with three fp16 MFMAs per product.  The two cross
terms are 2^-11 of the product; in fp8 (e4m3, twice the fp16 MFMA rate on gfx950) they would cost one MFMA-equivalent
instead of two.  This script EMULATES that arithmetic on the CPU, in float64 with exactly rounded operands, through the
whole eval forward of the oracle (every convolution and transposed convolution replaced), and reports the error of the
predicted frames against the exact float64 forward the way `bench.py`'s parity does (max |diff| / max |ref|):

    s16         hi, lo = 11-bit halves, three exact products                    (what the kernels do today)
    fp8x2       both cross terms on e4m3 operands, per-tensor power-of-two scale
    fp8x2-mx    ... with one power-of-two scale per 32 input channels (the MX block format of v_mfma_scale_*)
    fp8x1       only x_lo w_hi on e4m3, x_hi w_lo stays fp16                    (2.5 MFMA-equivalents)
    hi-only     no cross terms at all                                           (plain fp16 operands)

The memory lookups are forced to the exact forward's (oracle.quantize_topk force_idx), so that the numbers are
arithmetic, not re-routed lookups.  Test infrastructure: imports oracle/.

    python tools/fp8_cross_emulation.py [--size 64] [--batch 2] [--n-embed 256]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from ammcnet_aaai2021_amd import synthetic as S
from oracle import ammc_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--n-embed", type=int, default=256)
args = ap.parse_args()
torch.set_grad_enabled(False)


def round_sig(v, bits):
    """round to `bits` significant bits (round to nearest even), no range limit"""
    m, e = torch.frexp(v)
    return torch.ldexp(torch.round(m * 2.0 ** bits), e - bits)


def split16(v):
    hi = round_sig(v, 11)
    return hi, round_sig(v - hi, 11)


def e4m3(v, scale):
    """e4m3 (4 significant bits, normal exponents 2^-6 .. 2^8, subnormal quantum 2^-9, max 448) of v * scale, / scale"""
    u = (v * scale).clamp(-448.0, 448.0)
    _, e = torch.frexp(u)                                   # |u| in [2^(e-1), 2^e)
    q = torch.ldexp(torch.ones_like(u), torch.clamp(e - 4, min=-9))
    return torch.round(u / q) * q / scale


def pow2_scale(v, dims=None):
    a = v.abs().amax(dim=dims, keepdim=True) if dims is not None else v.abs().max()
    a = torch.clamp(a, min=1e-300)
    return torch.exp2(torch.floor(torch.log2(448.0 / a)))


def q8(v, mode, cdim):
    if mode == "tensor":
        return e4m3(v, pow2_scale(v))
    # one scale per block of 32 channels (dimension `cdim`) and per position / filter
    c = v.shape[cdim]
    pad = (-c) % 32
    shape = list(v.shape)
    vv = v
    if pad:
        pshape = list(shape)
        pshape[cdim] = pad
        vv = torch.cat([v, torch.zeros(pshape, dtype=v.dtype)], cdim)
    blk = list(vv.shape)
    blk[cdim:cdim + 1] = [vv.shape[cdim] // 32, 32]
    vb = vv.reshape(blk)
    out = e4m3(vb, pow2_scale(vb, dims=cdim + 1)).reshape(vv.shape)
    return out.narrow(cdim, 0, c)


MODE = ["exact"]


def emul(conv, x, w, b, **kw):
    mode = MODE[0]
    if mode == "exact":
        return conv(x, w, b, **kw)
    xh, xl = split16(x)
    wh, wl = split16(w)
    y = conv(xh, wh, b, **kw)
    wc = 0 if conv is F.conv2d else 0                        # conv2d: [out, in, kh, kw]; conv_transpose2d: [in, out, kh, kw]
    w_cdim = 1 if conv is F.conv2d else 0
    if mode == "hi-only":
        return y
    if mode == "s16":
        return y + conv(xh, wl, None, **kw) + conv(xl, wh, None, **kw)
    sm = "block" if mode.endswith("-mx") else "tensor"
    c2 = conv(q8(xl, sm, 1), q8(wh, sm, w_cdim), None, **kw)
    if mode.startswith("fp8x1"):
        return y + conv(xh, wl, None, **kw) + c2
    return y + conv(q8(xh, sm, 1), q8(wl, sm, w_cdim), None, **kw) + c2


_conv2d, _convt = F.conv2d, F.conv_transpose2d


def conv2d(x, w, b=None, stride=1, padding=0):
    return emul(_conv2d, x, w, b, stride=stride, padding=padding)


def convt(x, w, b=None, stride=1, padding=0):
    return emul(_convt, x, w, b, stride=stride, padding=padding)


O.F.conv2d, O.F.conv_transpose2d = conv2d, convt            # (O.F is torch.nn.functional: restored at exit)
try:
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in S.make_twostream_state(n_embed=args.n_embed).items()}
    rgb_x, op_x, _, _ = (t.double() for t in S.make_clips(args.batch, args.size, args.size, tag="fp8-emul"))
    MODE[0] = "exact"
    ref = O.twostream_forward(sd, rgb_x, op_x, 2, training=False, want_aux=True)
    idx = {p: ref[-1][f"{p}.idx"].reshape(-1, 2) for p in ("rgb", "op")}
    print(f"twostream eval forward, {args.batch} clips at {args.size}x{args.size}, {args.n_embed} slots; error of the predicted frames "
          f"against the exact float64 forward (max |diff| / max |ref|; north_star's gate: 1e-4)")
    for mode in ("s16", "fp8x1", "fp8x1-mx", "fp8x2", "fp8x2-mx", "hi-only"):
        MODE[0] = mode
        out = O.twostream_forward(sd, rgb_x, op_x, 2, training=False, force_idx=idx)
        errs = [float((out[i] - ref[i]).abs().max() / ref[i].abs().max()) for i in (0, 1)]
        l2 = [float((out[i] - ref[i]).norm() / ref[i].norm()) for i in (0, 1)]
        print(f"  {mode:9s} rgb {errs[0]:.2e}  flow {errs[1]:.2e}   (L2-relative {l2[0]:.2e} / {l2[1]:.2e})")
finally:
    O.F.conv2d, O.F.conv_transpose2d = _conv2d, _convt
