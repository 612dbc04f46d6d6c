"""Stand-alone forwards of the path's building blocks (NCHW in, NCHW out).

The whole-model forwards run on the cached plans of `engine.EvalEngine`; these
functions run ONE block on freshly allocated buffers through the same C-ABI
kernels, so that `double_conv(...)`, `down(...)`, `up(...)`, `bridge(...)`,
`Quantize_topk(...)` and the vq block work when called on their own, exactly
as the reference's sub-modules do, and so that every kernel has a per-op
parity test (tests/test_gpu_parity.py, tests/test_gpu_s16.py).  No ATen compute here either.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import ACT_NONE
from .engine import Act, Plan, _Builder, _DoubleConvPack, _Packer, _ptr


def _need_cuda(x: torch.Tensor):
    if not x.is_cuda:
        raise _lib.AmmcHipError("the HIP path needs tensors on the GPU; there is no CPU fallback")


def _stream(x) -> int:
    return torch.cuda.current_stream(x.device).cuda_stream


def _to_halo(bld: _Builder, x: torch.Tensor, cp: int | None = None, into: Act | None = None) -> Act:
    """NCHW tensor -> halo-padded NHWC activation (optionally a channel slice of a wider buffer)"""
    B, C, H, W = x.shape
    cp = cp or (C + 3) // 4 * 4
    a = into if into is not None else bld.act(B, H, W, cp)
    x = x.detach().float().contiguous()
    bld.plan.keep.append(x)
    bld.plan.add(bld.lib.ammc_nchw_to_nhwc_f32, _ptr(x), B, C, H, W, a.pix0(), *a.strides, cp, name="to_nhwc")
    return a


def _to_nchw(bld: _Builder, a: Act) -> torch.Tensor:
    y = torch.empty((a.B, a.c, a.H, a.W), device=a.buf.device, dtype=torch.float32)
    bld.plan.add(bld.lib.ammc_nhwc_to_nchw_f32, a.pix0(), *a.strides, a.B, a.c, a.H, a.W, _ptr(y), name="to_nchw")
    return y


def double_conv_eval(dc, x: torch.Tensor, pool_first: bool = False, residual: torch.Tensor | None = None):
    """`double_conv.forward` / `down.forward` / one half of `bridge.forward` in eval mode"""
    _need_cuda(x)
    plan = Plan()
    bld = _Builder(plan, x.device)
    pack = _DoubleConvPack(_Packer(x.device), dc)
    a = _to_halo(bld, x, pack.cin_p)
    if pool_first:
        p = bld.act(a.B, a.H // 2, a.W // 2, a.c)
        bld.maxpool(a, p)
        a = p
    mid = bld.act(a.B, a.H, a.W, pack.cout)
    out = bld.act(a.B, a.H, a.W, pack.cout)
    res = _to_halo(bld, residual) if residual is not None else None
    bld.double_conv(a, pack, mid, out, res=res)
    y = _to_nchw(bld, out)
    plan.run(_stream(x))
    return y


def up_eval(upm, x1: torch.Tensor, x2: torch.Tensor):
    """`up.forward`: ConvTranspose into the second half of the concat buffer, skip into the first"""
    _need_cuda(x1)
    dy, dx = x2.shape[2] - 2 * x1.shape[2], x2.shape[3] - 2 * x1.shape[3]
    if dy not in (0, 1) or dx not in (0, 1):
        raise ValueError("up: the skip tensor must be 2x the upsampled one, or one row / column more (what MaxPool2d(2) "
                         "leaves of an odd level; `up.forward` pads that row / column with zeros, unet.py:53-56)")
    plan = Plan()
    bld = _Builder(plan, x1.device)
    pk = _Packer(x1.device)
    c = x2.shape[1]
    B, _, H, W = x2.shape
    cat = bld.act(B, H, W, 2 * c)
    _to_halo(bld, x2, into=cat.slice(0, c))          # the skip tensor: channels [0, c)
    a = _to_halo(bld, x1)
    bld.convt(a, pk.convt(upm.up.weight), upm.up.bias.detach(), cat.slice(c, c))
    pack = _DoubleConvPack(pk, upm.conv)
    mid = bld.act(B, H, W, pack.cout)
    out = bld.act(B, H, W, pack.cout)
    bld.double_conv(cat, pack, mid, out)
    y = _to_nchw(bld, out)
    plan.run(_stream(x1))
    return y


def quantize_topk_eval(qmod, x: torch.Tensor):
    """`Quantize_topk.forward` in eval mode on an NHWC tensor [B,h,w,D]"""
    _need_cuda(x)
    lib = _lib.load()
    B, h, w, d = x.shape
    n, m, k = B * h * w, qmod.n_embed, qmod.k
    x = x.detach().float().contiguous()
    e_md, enorm = _Packer(x.device).codebook(qmod.embed)
    idx = torch.empty((n, k), device=x.device, dtype=torch.int32)
    qk = torch.empty((B, h, w, k * d), device=x.device, dtype=torch.float32)
    q1 = torch.empty((B, h, w, d), device=x.device, dtype=torch.float32)
    nblk = lib.ammc_memory_topk_blocks(n)
    part = torch.empty(nblk, device=x.device, dtype=torch.float32)
    diff = torch.empty(1, device=x.device, dtype=torch.float32)
    s = _stream(x)
    _lib.check(lib.ammc_memory_topk_fwd_f32(_ptr(x), _ptr(qmod.embed), _ptr(e_md), _ptr(enorm), n, d, m, k,
                                            idx.data_ptr(), _ptr(qk), _ptr(q1), _ptr(part), s), "memory_topk")
    _lib.check(lib.ammc_sum_partials_f32(_ptr(part), nblk, 1.0 / float(n * d), _ptr(diff), s), "sum_partials")
    qmod.last_indices = idx.view(B, h, w, k)
    return qk, diff[0], q1


def vq_block_eval(quan, x: torch.Tensor, residual: bool):
    """`enc_quan_dec_topk.forward` (+ `out += x` of enc_quan_dec_res_topk) in eval mode"""
    _need_cuda(x)
    plan = Plan()
    bld = _Builder(plan, x.device)
    pk = _Packer(x.device)
    lib = bld.lib
    a = _to_halo(bld, x)
    B, H, W = a.B, a.H, a.W
    q = quan.quantize
    d, m, k = q.dim, q.n_embed, q.k
    n = B * H * W
    enc_w, _ = pk.conv(quan.enc.weight, 1)
    dec_w, _ = pk.conv(quan.dec.weight, 1)
    e_md, enorm = pk.codebook(q.embed)
    z = bld.act(B, H, W, d, halo=0)
    bld.conv(a, enc_w, z, ntaps=1, cin=a.c, n=d, shift=quan.enc.bias.detach(), name="enc")
    idx = torch.empty((n, k), device=x.device, dtype=torch.int32)
    qk = bld.act(B, H, W, k * d, halo=0)
    q1 = bld.buf(B, H, W, d)
    nblk = lib.ammc_memory_topk_blocks(n)
    part = bld.buf(nblk)
    diff = bld.buf(1)
    plan.keep.extend([idx, e_md, enorm])
    plan.add(lib.ammc_memory_topk_fwd_f32, _ptr(z.buf), _ptr(q.embed), _ptr(e_md), _ptr(enorm), n, d, m, k,
             idx.data_ptr(), _ptr(qk.buf), _ptr(q1), _ptr(part), name="memory_topk")
    plan.add(lib.ammc_sum_partials_f32, _ptr(part), nblk, 1.0 / float(n * d), _ptr(diff), name="diff")
    out = bld.act(B, H, W, a.c)
    bld.conv(qk, dec_w, out, ntaps=1, cin=k * d, n=a.c, shift=quan.dec.bias.detach(),
             res=a if residual else None, act=ACT_NONE, name="dec")
    y = _to_nchw(bld, out)
    plan.run(_stream(x))
    return y, diff, q1


def embed_rows(embed: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """`Quantize_topk.embed_code` (unet.py:315-316): rows of the transposed codebook"""
    return embed.t()[ids]


# which kernel `quantize_topk_f16` / `workload.MemoryStress` launch: 1 (default) = rows resident in registers
# (csrc/memory_topk_f16r.hip, round 5), 0 = the round-2 form (csrc/memory_topk_f16.hip: rows in LDS, codebook L2 -> registers)
F16_ROWS_IN_REGISTERS = os.environ.get("AMMC_MEMORY_F16_FORM", "1") != "0"


def quantize_topk_f16(embed: torch.Tensor, x: torch.Tensor, k: int):
    """Stress form of the memory addressing (BASELINE.json config 5): fp16 MFMA operands, fp32
    accumulation.  embed [D,M] fp32, x [..., D] fp32 -> (q_topk [..., k*D], diff, q_one, idx [..., k]).
    Not the parity path (the slot ranking sees fp16-rounded operands)."""
    _need_cuda(x)
    lib = _lib.load()
    d, m = embed.shape
    lead = x.shape[:-1]
    x2 = x.detach().float().contiguous().view(-1, d)
    n = x2.shape[0]
    dev = x.device
    s = _stream(x)
    idx = torch.empty((n, k), device=dev, dtype=torch.int32)
    qk = torch.empty((n, k * d), device=dev, dtype=torch.float32)
    q1 = torch.empty((n, d), device=dev, dtype=torch.float32)
    diff = torch.empty(1, device=dev, dtype=torch.float32)
    e_md, _ = _Packer(dev).codebook(embed)
    if F16_ROWS_IN_REGISTERS:
        # round 5: feature rows resident in registers, codebook tiles through LDS (csrc/memory_topk_f16r.hip)
        tiles = torch.empty(lib.ammc_codebook_f16_tiles_bytes(d, m), device=dev, dtype=torch.uint8)
        _lib.check(lib.ammc_pack_codebook_f16_tiles(_ptr(embed), d, m, tiles.data_ptr(), s), "pack_codebook_f16_tiles")
        nblk = lib.ammc_memory_topk_f16r_blocks(n)
        part = torch.empty(nblk, device=dev, dtype=torch.float32)
        _lib.check(lib.ammc_memory_topk_fwd_f16r(_ptr(x2), tiles.data_ptr(), _ptr(e_md), n, d, m, k, idx.data_ptr(), _ptr(qk),
                                                 _ptr(q1), _ptr(part), s), "memory_topk_f16r")
    else:
        mpad = (m + 31) // 32 * 32
        e_kblk = torch.empty((d // 8, mpad, 8), device=dev, dtype=torch.float16)
        enorm16 = torch.empty(m, device=dev, dtype=torch.float32)
        _lib.check(lib.ammc_pack_codebook_f16(_ptr(embed), d, m, e_kblk.data_ptr(), _ptr(enorm16), s), "pack_codebook_f16")
        nblk = lib.ammc_memory_topk_f16_blocks(n)
        part = torch.empty(nblk, device=dev, dtype=torch.float32)
        _lib.check(lib.ammc_memory_topk_fwd_f16(_ptr(x2), e_kblk.data_ptr(), _ptr(e_md), _ptr(enorm16), n, d, m, k,
                                                idx.data_ptr(), _ptr(qk), _ptr(q1), _ptr(part), s), "memory_topk_f16")
    _lib.check(lib.ammc_sum_partials_f32(_ptr(part), nblk, 1.0 / float(n * d), _ptr(diff), s), "sum_partials")
    return qk.view(*lead, k * d), diff[0], q1.view(*lead, d), idx.view(*lead, k)
