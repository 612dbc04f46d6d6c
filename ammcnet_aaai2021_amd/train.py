"""Training-mode forward and backward of the path on the HIP kernels.

`TrainEngine.forward` runs the network in `.train()` semantics (batch-statistics BatchNorm
with running-stat update, EMA codebook update, reference unet.py:298-309) and keeps what the
backward needs in a per-shape workspace; `TrainEngine.backward` is the hand-scheduled reverse
pass (the reference gets it from torch.autograd): input-gradient convolutions run through
`ammc_conv_gemm_f32` with flipped/transposed filters, weight gradients through
`ammc_conv_wgrad_f32`, everything else through the streaming kernels of train_kernels.hip.
`HipPathFunction` surfaces the pair as ONE torch.autograd.Function whose inputs are the clips
and all parameters, so optimizers / DDP / RCCL all-reduce see ordinary `.grad` tensors.

Gradient flow facts this file relies on: the memory read is a lookup into a
buffer, so the only gradient reaching `enc` is the commit term; `dec` gets ordinary GEMM
gradients; the vq residual passes the gradient through unchanged.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_TANH, AmmcConvDesc, AmmcWgradDesc
from .engine import Act, _cin_pad, _kpad, _ptr, s16_variant

BN_MOMENTUM = 0.1
CHANS = (64, 128, 256, 512)
# 3x3 convolutions of the training path (forward and input-gradient): "s16" = the fp32-equivalent split-fp16 kernels
# of the inference path (operands converted per launch, fp32 outputs; everything else - statistics, weight gradients,
# 1x1 / transposed convs - stays on the fp32 kernels), "fp32" = exact fp32 MFMA throughout
# Process-wide default; per model: `model.train_precision = "fp32" | "s16"` (read when the engine is built / rebuilt)
TRAIN_PRECISION = os.environ.get("AMMC_TRAIN_PRECISION", "s16")
WGRAD_S16 = os.environ.get("AMMC_WGRAD_S16", "1") != "0"          # the 3x3 weight gradients as well (wgrad_s16.hip)
# ConvTranspose forward / input gradient on the S16 kernels too: opt-in - its short-K GEMMs gain less than the operand
# re-encoding costs (74.9 ms/step without, 75.5 with)
CONVT_S16 = os.environ.get("AMMC_CONVT_S16", "0") != "0"          # ConvTranspose input gradient too (re-encodes its operand)
# Round 4: every activation that only split-fp16 kernels read again is written as its S16 twin BY ITS PRODUCER (the
# BatchNorm apply pass stores the fp32 tensor and the twin in one sweep, the max-pool runs on twins, the ConvTranspose
# forward runs on conv_gemm_s16 with an S16 output): no fp32 -> S16 re-encoding pass of an activation is left in the
# forward (they were 2.4 ms of the 70-ms step), and the ConvTranspose forward leaves the fp32 MFMA pipe.
TWIN_S16 = os.environ.get("AMMC_TWIN_S16", "1") != "0"
# ... and in the backward the gradient of a ConvTranspose's output is re-encoded ONCE (its max |g| comes out of the
# bias-gradient pass that reads it anyway, the re-encoding touches its half of the concat buffer only) for BOTH of the
# layer's gradient kernels: weight gradient on ammc_conv_wgrad_s16 (2x2 window, stride 2), input gradient on
# ammc_conv_gemm_s16.  (AMMC_CONVT_S16 alone re-encoded the whole concat buffer for the input gradient only: no gain.)
CONVT_GRADS_S16 = os.environ.get("AMMC_CONVT_GRADS_S16", "1") != "0"
# all 3x3 filters of a step packed to their S16 images by ONE launch per direction (ammc_pack_filters_s16) instead of a
# pack and a split launch per layer and direction (~140 launches of 5-8 us per step)
PACK_BATCH = os.environ.get("AMMC_PACK_BATCH", "1") != "0"
# BatchNorm batch statistics as a second output of the convolution that produces the tensor (AmmcConvDesc.stats: one
# partial row per 8 x 32 output patch, from the accumulators) instead of a pass that re-reads it
FUSE_BN_STATS = os.environ.get("AMMC_FUSE_BN_STATS", "1") != "0"
STAT_SEG = 128
# max-pool backward from the window positions recorded by the forward (a byte per pooled element) instead of finding the
# maxima again from the pooled tensor's S16 twin
POOL_IDX = os.environ.get("AMMC_POOL_IDX", "1") != "0"
# ... and the pooled tensor + those positions as extra outputs of the BatchNorm apply pass that writes the tensor
# (ammc_scale_shift_act_s16_pool_f32) instead of a max-pool pass that reads it back
FUSE_POOL_APPLY = os.environ.get("AMMC_FUSE_POOL_APPLY", "1") != "0"
# ... and in the backward the gradient of a pooled tensor (skip gradient + max-pool backward) is never materialised: the
# BatchNorm-backward passes of the unit that produced the tensor form it on the fly from those positions
FUSE_UNPOOL_BN = os.environ.get("AMMC_FUSE_UNPOOL_BN", "1") != "0"
# the reduction of a unit's BatchNorm backward (sum g, sum g xhat, max |g|, max |xhat|) as a second output of the
# input-gradient convolution that PRODUCES the unit's output gradient (AmmcConvDesc.bn_c), instead of a pass that reads
# the gradient back
FUSE_BN_BWD_STATS = os.environ.get("AMMC_FUSE_BN_BWD_STATS", "1") != "0"
# one rank: the rgb and the flow stream of the network (independent between the bridge / memory joints) are issued on two
# HIP streams, so that one's HBM-bound BatchNorm passes (~1170 W at 6 TB/s) run beside the other's MFMA-bound convolutions
# instead of after them: the step is energy-bound and those passes leave ~230 W of the 1400 W cap unused
TWO_STREAMS = os.environ.get("AMMC_TWO_STREAMS", "1") != "0"
# the 3x3 weight gradients' split partials as slabs summed straight into the parameter gradient (round 5) instead of fp32
# atomics into a zeroed packed buffer + an unpack launch
WGRAD_SLABS = os.environ.get("AMMC_WGRAD_SLABS", "1") != "0"
MID_S16 = os.environ.get("AMMC_MID_S16", "1") != "0"              # double_conv middle activations exist as S16 only
FUSE_BN_BWD = os.environ.get("AMMC_FUSE_BN_BWD", "1") != "0"      # BN backward writes the S16 twin of dc (one rank)


class _WS:
    """workspace allocator: zero-initialised once (kernels only write interiors)"""

    ZCHUNK = 16 << 20          # floats per chunk of the per-step-zeroed slab (64 MB)

    def __init__(self, device, zchunk: Optional[int] = None):
        self.device = device
        self.lib = _lib.load()
        self.bytes = 0
        if zchunk is not None:                  # stand-alone block engines: a handful of small accumulators
            self.ZCHUNK = int(zchunk)
        self._zchunks: List[torch.Tensor] = []
        self._zfree = 0

    def buf(self, *shape, dtype=torch.float32) -> torch.Tensor:
        t = torch.zeros(shape, device=self.device, dtype=dtype)
        self.bytes += t.numel() * t.element_size()
        return t

    def zbuf(self, *shape, dtype=torch.float32) -> torch.Tensor:
        """a buffer that must be ZERO at the start of every backward (weight-gradient accumulators, the slots of the
        max-|g| reductions): carved out of a few large chunks that `zero_step` clears with one memset each, instead
        of one fill kernel per buffer per step (there are ~180 of them)"""
        n = 1
        for d in shape:
            n *= d
        n_al = (n + 63) // 64 * 64                            # 256-byte granules
        if not self._zchunks or self._zfree + n_al > self._zchunks[-1].numel():
            self._zchunks.append(self.buf(max(self.ZCHUNK, n_al)))
            self._zfree = 0
        flat = self._zchunks[-1][self._zfree:self._zfree + n]
        self._zfree += n_al
        t = flat.view(dtype).view(*shape) if dtype != torch.float32 else flat.view(*shape)
        t._ammc_zslab = True
        return t

    def zero_step(self) -> None:
        for c in self._zchunks:
            c.zero_()

    def act(self, B, H, W, c, halo=1) -> Act:
        return Act(self.buf(B, H + 2 * halo, W + 2 * halo, c), B, H, W, c, 0, halo)


def _check_embed_dim(d: int) -> None:
    """the 1x1 `enc` weight / input gradients read the [rows][d] gradient of z in 32-channel tiles: a narrower or
    ragged row would make them read the neighbouring pixels' values (and past the buffer at the last pixel)"""
    if d % 32:
        raise NotImplementedError(f"memory block in training mode: embed_dim {d} (multiples of 32; the reference uses 64)")


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _chk(rc, what):
    if rc != 0:
        _lib.check(rc, what)


class _Ops:
    """immediate-mode launches of the C ABI on the current stream"""

    def __init__(self, ws: _WS, precision: Optional[str] = None):
        self.ws, self.lib, self.dev = ws, ws.lib, ws.device
        self.zeros = ws.buf(1024)
        self.sync_group = None          # set per step by TrainEngine: a process group, or False for "no sync"
        self.sync_force = False         # parallel.sync_statistics(..., force=True): the collective path with one rank too
        self.s16 = (precision or TRAIN_PRECISION) == "s16"
        self._shadows: Dict[int, torch.Tensor] = {}
        self.amax = ws.buf(256, dtype=torch.int32)         # slots of ammc_absmax_bits_f32 / bn_bwd_apply
        self.side = None                                   # two HIP streams of `_side_by_side`, made on first use
        self._wgrad_slabs: Dict[int, torch.Tensor] = {}    # slab workspace of the 3x3 weight gradients, per HIP stream
        # bench.py: a list here brackets every MFMA launch of the 3x3 layers with HIP events on the launch stream and
        # collects (kernel label, algorithmic flops, start event, end event)
        self.timing: Optional[list] = None
        self.collectives = 0            # statistics all-reduces issued (tests: the streams share them)
        self.pack_items = {"fwd": [], "bwd": []}      # (weight, out16, cout, cin, inner_p, kpad, kind, rows) of every 3x3 unit
        self._pack_tables: Dict[str, tuple] = {}
        self.packed = {"fwd": False, "bwd": False}    # this step's batched pack has run (TrainEngine sets / clears)
        self.nbt: Optional[list] = None  # TrainEngine.forward: the BatchNorm step counters of this forward (32 one-element adds -> one launch)

    def _mfma_launch(self, label, flops: float, call, what: str):
        if self.timing is None:
            _chk(call(), what)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _chk(call(), what)
        e1.record()
        self.timing.append((label() if callable(label) else label, flops, e0, e1))

    def run_pack(self, which: str) -> None:
        """every registered 3x3 filter of one direction -> its S16 image, one launch.  The device table is rebuilt when a
        parameter's storage moved (`.to()`, re-assignment); in-place updates (optimizers, load_state_dict) keep it."""
        items = self.pack_items[which]
        if not items:
            return
        import numpy as np
        ptrs = tuple(it[0].data_ptr() for it in items)
        tab = self._pack_tables.get(which)
        if tab is None or tab[0] != ptrs:
            dt = np.dtype([("w", "u8"), ("out", "u8"), ("cout", "i4"), ("cin", "i4"), ("inner_p", "i4"), ("kpad", "i4"),
                           ("kind", "i4"), ("rows", "i4"), ("group_end", "i8")])
            assert dt.itemsize == self.lib.ammc_pack_filters_item_bytes()
            arr = np.zeros(len(items), dtype=dt)
            total = 0
            for i, (w, out16, cout, cin, inner_p, kpad, kind, rows) in enumerate(items):
                total += rows * kpad // 8
                arr[i] = (w.data_ptr(), out16.data_ptr(), cout, cin, inner_p, kpad, kind, rows, total)
            dev = torch.from_numpy(arr.view(np.uint8).copy()).to(self.dev)
            tab = (ptrs, dev, len(items), total)
            self._pack_tables[which] = tab
        _chk(self.lib.ammc_pack_filters_s16(tab[1].data_ptr(), tab[2], tab[3], self.s), f"pack_filters({which})")
        self.packed[which] = True

    def shadow(self, a: Act) -> Act:
        """the S16 twin of an fp32 NHWC buffer (same geometry, allocated once per underlying buffer)"""
        key = a.buf.data_ptr()
        buf = self._shadows.get(key)
        if buf is None:
            buf = self._shadows[key] = self.ws.buf(*a.buf.shape)
        t = Act(buf, a.B, a.H, a.W, a.c, a.c_off, a.halo)
        t.rs, t.bs = a.rs, a.bs                       # (a cropped view keeps its buffer's strides)
        return t

    def to_s16(self, x: Act, rescale: bool = False, have_amax: bool = False, amax: Optional[torch.Tensor] = None):
        """re-encode the fp32 buffer behind `x` into its S16 twin; `rescale` (gradients): first bring it into the half
        range by a power of two found on the device (`have_amax`: the producer already left max |x| in the slots;
        `amax`: the caller's own slots, zero at the start of the backward pass - else the shared ones, cleared here).
        Returns (twin, inverse scale [1] or None)."""
        lib, s = self.lib, self.s
        xs = self.shadow(x)
        inv = None
        if rescale:
            inv = torch.empty(1024, device=self.dev, dtype=torch.float32)
            if amax is None:
                # the SHARED slots: one user at a time.  Inside `_side_by_side` two generators run on two HIP streams at
                # once and each must bring its own slots (the units do: per-unit amax buffers) - a caller that forgot
                # would race silently, so it is refused here
                if self.side is not None and torch.cuda.current_stream(self.dev) in self.side:
                    raise RuntimeError("to_s16(rescale=True) without its own `amax` slots on a side stream of "
                                       "_side_by_side: the shared slots would be raced by the other stream")
                amax = self.amax
                if not have_amax:
                    amax.zero_()
            if not have_amax:
                _chk(lib.ammc_absmax_bits_f32(_ptr(x.buf), x.buf.numel(), amax.data_ptr(), s), "absmax")
            _chk(lib.ammc_split_rows_scaled_f32(_ptr(x.buf), x.buf.numel(), _ptr(xs.buf), amax.data_ptr(), _ptr(inv),
                                                1024, s), "split_rows_scaled(x)")
        else:
            _chk(lib.ammc_split_rows_f32(_ptr(x.buf), x.buf.numel(), _ptr(xs.buf), s), "split_rows(x)")
        return xs, inv

    def wgrad_s16(self, g16: Act, a16: Act, dw: torch.Tensor, inv, *, n, cin, what="wgrad", true_nc=None, ntaps=9, a_step=1,
                  out_oihw: Optional[torch.Tensor] = None):
        """weight gradient from the S16 twins of the output gradient and of the layer input (3x3; ntaps 4 / a_step 2: the
        ConvTranspose form of ammc_conv_wgrad_f32 - g = the layer input, a = the output gradient at twice the resolution).
        `out_oihw` (3x3 layers): the parameter's gradient tensor; where the layer's kernel has a slab form (round 5,
        AMMC_WGRAD_SLABS) the split partials are stored as slabs and summed straight into it - no atomics into the zeroed
        packed buffer `dw`, no unpack launch - and the call returns True (the caller then skips its unpack)."""
        d = AmmcWgradDesc()
        d.g, d.a, d.dw, d.zeros = g16.pix0(), (a16.tap0() if ntaps == 9 else a16.pix0()), _ptr(dw), _ptr(self.zeros)
        d.batch, d.height, d.width = g16.B, g16.H, g16.W
        d.n, d.cin, d.ntaps, d.a_step = n, cin, ntaps, a_step
        d.g_bs, d.g_rs, d.g_ps = g16.strides
        d.a_bs, d.a_rs, d.a_ps = a16.strides
        label = ("conv_wgrad_s16 (3x3 weight gradients: wgrad_tap3_s16 / wgrad_tap_s16 instances)" if ntaps == 9 else
                 "conv_wgrad_s16 (ConvTranspose weight gradients: wgrad_s16)")
        flops = 2.0 * g16.B * g16.H * g16.W * ntaps * (true_nc if true_nc is not None else n * cin)      # unpadded channels
        if out_oihw is not None and ntaps == 9 and a_step == 1 and WGRAD_SLABS:
            need = int(self.lib.ammc_conv_wgrad_s16_slab_floats(C.byref(d)))
            if need > 0:
                # one slab workspace per HIP stream (the rgb / flow halves of a step run on two): sized for the largest user
                key = torch.cuda.current_stream(self.dev).cuda_stream
                ws = self._wgrad_slabs.get(key)
                if ws is None or ws.numel() < need:
                    ws = self._wgrad_slabs[key] = torch.empty(max(need, 20 << 20), device=self.dev, dtype=torch.float32)
                cout, cin_t = out_oihw.shape[0], out_oihw.shape[1]
                self._mfma_launch(label, flops, lambda: self.lib.ammc_conv_wgrad_s16_slabs(
                    C.byref(d), _ptr(inv) if inv is not None else None, _ptr(ws), ws.numel(), _ptr(out_oihw), cout, cin_t, self.s), what)
                return True
        if not getattr(dw, "_ammc_zslab", False):          # slab buffers were cleared by `_WS.zero_step` (backward start)
            dw.zero_()
        self._mfma_launch(label, flops,
                          lambda: self.lib.ammc_conv_wgrad_s16(C.byref(d), _ptr(inv) if inv is not None else None, self.s), what)
        return False

    def conv_s16(self, x: Act, w: torch.Tensor, y: Act, *, ntaps, cin, n, res: Optional[Act] = None, what="conv",
                 rescale: bool = False, pre=None, shift=None, up=1, cgroup=None, x_step=1, y_s16: bool = False,
                 w16: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, bn=None):
        """3x3 conv on the split-fp16 MFMA kernels: x (fp32, any channel slice of its buffer) is re-encoded into its S16
        twin (or `pre` = what `to_s16` returned for it), the packed filter likewise; fp32 output (+ fp32 residual), or -
        `y_s16`: y is the S16 twin itself - an S16 output.  ammc_conv_gemm_s16 picks the kernel."""
        lib, s = self.lib, self.s
        xs, inv = pre if pre is not None else self.to_s16(x, rescale)
        if w16 is None:                              # (else: the S16 image is already there - the step's batched pack)
            w16 = torch.empty_like(w)
            _chk(lib.ammc_split_rows_f32(_ptr(w), w.numel(), _ptr(w16), s), "split_rows(w)")
        d = AmmcConvDesc()
        d.x = xs.tap0() if ntaps == 9 else xs.pix0()
        d.w, d.y = _ptr(w16), y.pix0()
        d.res = res.pix0() if res is not None else None
        d.scale = _ptr(inv) if inv is not None else None
        d.shift = _ptr(shift) if shift is not None else None
        d.batch, d.height, d.width = y.B, y.H // up, y.W // up
        d.cin, d.ntaps, d.n, d.up, d.act, d.y_f32, d.x_step = cin, ntaps, n, up, ACT_NONE, 0 if y_s16 else 1, x_step
        d.cgroup = cgroup if cgroup is not None else n
        d.x_bs, d.x_rs, d.x_ps = xs.strides
        d.y_bs, d.y_rs, d.y_ps = y.strides
        if res is not None:
            d.r_bs, d.r_rs, d.r_ps = res.strides
        d.stats = _ptr(stats) if stats is not None else None
        if bn is not None:                           # `stats` = the BatchNorm-backward partial rows of unit `bn`, which y goes to
            d.bn_c = bn.craw.pix0()
            d.bn_bs, d.bn_rs, d.bn_ps = bn.craw.strides
            d.bn_mean, d.bn_invstd, d.bn_scale, d.bn_shift, d.bn_relu = _ptr(bn.mean), _ptr(bn.invstd), _ptr(bn.scale), _ptr(bn.shift), 1
        # (fp32 outputs: no S16 range flag here - an operand beyond the half range becomes inf in `to_s16` and the
        # loss turns non-finite, which the training loop sees)
        self._mfma_launch(lambda: s16_variant(d).replace("+stats", "").replace("+bnbwd", ""), 2.0 * d.batch * d.height * d.width * ntaps * cin * n,
                          lambda: lib.ammc_conv_gemm_s16(C.byref(d), s), what)

    def conv_s16_stats_rows(self, x: Act, w: torch.Tensor, y: Act, *, cin, n, bn=None) -> int:
        """rows of the statistics output (`AmmcConvDesc.stats`) of the kernel `conv_s16` would launch for this 3x3 layer
        with an fp32 output, 0 when that kernel has no statistics epilogue (`bn`: the BatchNorm-backward form for unit bn)"""
        d = AmmcConvDesc()
        if bn is not None:
            d.bn_c = bn.craw.pix0()
            d.bn_bs, d.bn_rs, d.bn_ps = bn.craw.strides
            d.bn_mean, d.bn_invstd, d.bn_scale, d.bn_shift, d.bn_relu = _ptr(bn.mean), _ptr(bn.invstd), _ptr(bn.scale), _ptr(bn.shift), 1
        d.x, d.w, d.y = x.tap0(), _ptr(w), y.pix0()           # (addresses are only checked for alignment)
        d.batch, d.height, d.width = y.B, y.H, y.W
        d.cin, d.ntaps, d.n, d.up, d.act, d.y_f32, d.x_step, d.cgroup = cin, 9, n, 1, ACT_NONE, 1, 1, n
        d.x_bs, d.x_rs, d.x_ps = x.strides
        d.y_bs, d.y_rs, d.y_ps = y.strides
        return int(self.lib.ammc_conv_gemm_s16_stats_rows(C.byref(d)))

    @property
    def sync_world(self) -> int:
        """> 1 when BatchNorm statistics / EMA counts are to be all-reduced across ranks (parallel.sync_statistics)"""
        import torch.distributed as dist
        if self.sync_group is False or not dist.is_initialized():
            return 1
        return dist.get_world_size(self.sync_group)

    @property
    def sync_on(self) -> bool:
        """the units take their all-reduce path: more than one rank, or one rank with `force` (the one-GPU RCCL test)"""
        import torch.distributed as dist
        return self.sync_world > 1 or (self.sync_force and self.sync_group is not False and dist.is_initialized())

    def all_reduce(self, t: torch.Tensor) -> None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.sync_group)

    @property
    def s(self):
        return _stream(self.dev)

    def conv(self, x: Act, w: torch.Tensor, y: Act, *, ntaps, cin, n, scale=None, shift=None, act=ACT_NONE,
             res: Optional[Act] = None, up=1, cgroup=None, x_step=1, what="conv"):
        d = AmmcConvDesc()
        d.x = x.tap0() if ntaps == 9 else x.pix0()
        d.w, d.y = _ptr(w), y.pix0()
        d.scale = _ptr(scale) if scale is not None else None
        d.shift = _ptr(shift) if shift is not None else None
        d.res = res.pix0() if res is not None else None
        d.batch, d.height, d.width = y.B // 1, (y.H // up), (y.W // up)
        d.cin, d.ntaps, d.n, d.up, d.act, d.x_step = cin, ntaps, n, up, act, x_step
        d.cgroup = cgroup if cgroup is not None else n
        d.x_bs, d.x_rs, d.x_ps = x.strides
        d.y_bs, d.y_rs, d.y_ps = y.strides
        if res is not None:
            d.r_bs, d.r_rs, d.r_ps = res.strides
        _chk(self.lib.ammc_conv_gemm_f32(C.byref(d), self.s), what)

    def wgrad(self, g: Act, a: Act, dw: torch.Tensor, *, n, cin, ntaps, a_step=1, what="wgrad"):
        if not getattr(dw, "_ammc_zslab", False):
            dw.zero_()
        d = AmmcWgradDesc()
        d.g = g.pix0()
        d.a = a.tap0() if ntaps == 9 else a.pix0()
        d.dw, d.zeros = _ptr(dw), _ptr(self.zeros)
        d.batch, d.height, d.width = g.B, g.H, g.W
        d.n, d.cin, d.ntaps, d.a_step = n, cin, ntaps, a_step
        d.g_bs, d.g_rs, d.g_ps = g.strides
        d.a_bs, d.a_rs, d.a_ps = a.strides
        _chk(self.lib.ammc_conv_wgrad_f32(C.byref(d), self.s), what)

    def chan_sum(self, x: Act, c: int, scratch: torch.Tensor) -> torch.Tensor:
        """per-channel sum over pixels of the first c channels of x (bias gradient)"""
        nb = self.lib.ammc_chan_reduce_blocks(x.B * x.H * x.W)
        _chk(self.lib.ammc_chan_sum_f32(x.pix0(), *x.strides, x.B, x.H, x.W, c, _ptr(scratch), self.s), "chan_sum")
        out = torch.empty(c, device=self.dev, dtype=torch.float32)
        _chk(self.lib.ammc_reduce_partials_f32(_ptr(scratch), nb, c, 1.0, _ptr(out), self.s), "reduce_partials")
        return out


def _side_by_side(ops: "_Ops", grads, *gens) -> bool:
    """One rank, nothing to all-reduce: run the generators - the rgb and the flow stream's share of a phase, or the two
    halves of the bridge: disjoint buffers - each on a HIP stream of its own, forked from the caller's stream and joined
    into it.  `ops.s` is the current stream, so the launches and the temporaries of a generator follow it there; the
    gradients it leaves in `grads` outlive the join and are consumed on the caller's stream, which the caching allocator
    is told (`record_stream`).  Returns False (nothing run) where this form does not apply: the caller falls back to
    `_lockstep`."""
    if not TWO_STREAMS or len(gens) != 2 or ops.sync_on or ops.timing is not None:
        return False
    cur = torch.cuda.current_stream(ops.dev)
    if ops.side is None:
        ops.side = (torch.cuda.Stream(ops.dev), torch.cuda.Stream(ops.dev))
    before = set(id(k) for k in grads) if grads is not None else set()
    try:
        # (round 6: the second stream deliberately a kernel or so BEHIND the first - so that one's convolution meets the
        # other's BatchNorm pass instead of its twin - was measured with a 200 / 500 / 1000 / 2000 us head start per
        # phase: 58.98 / 59.12 / 60.2 / 62.4 ms against 59.09 / 58.87 without.  Nothing; the two streams drift apart by
        # themselves.)
        for st, g in zip(ops.side, gens):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for _ in g:
                    raise RuntimeError("a generator asked for a collective on the one-rank path")
    finally:
        # always joined, also when a generator raised mid-phase (a refused launch, the RuntimeError above): work is then
        # still in flight on the side streams, on workspace buffers that whatever the caller runs next - a retry on the
        # fp32 kernels, the next step - re-uses on ITS stream
        for st in ops.side:
            cur.wait_stream(st)
    if grads is not None:
        for k, v in grads.items():
            if id(k) not in before:
                v.record_stream(cur)
    return True


def _lockstep(ops: "_Ops", *gens) -> None:
    """Advance the generators together.  A unit that needs statistics summed over the ranks (`parallel.sync_statistics`)
    yields the tensor(s) right where the reference-order code would all-reduce them; whatever the generators yield in
    the same round - the same layer of the rgb and of the flow stream, the two halves of the bridge, the four EMA
    tensors of the two memories - travels as ONE flat collective and is written back in place.  (Within one stream
    consecutive layers cannot share a collective: BatchNorm's statistics are needed before the next conv can run.)
    Without synchronised statistics nothing yields and the generators just run to completion, one after the other."""
    live = list(gens)
    while live:
        pend, nxt = [], []
        for g in live:
            try:
                t = next(g)
            except StopIteration:
                continue
            pend.extend(t if isinstance(t, (list, tuple)) else [t])
            nxt.append(g)
        if len(pend) == 1:
            ops.all_reduce(pend[0])
        elif pend:
            flat = torch.cat([t.reshape(-1) for t in pend])
            ops.all_reduce(flat)
            off = 0
            for t in pend:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        ops.collectives += 1 if pend else 0
        live = nxt


class _ConvBN:
    """conv3x3 (no bias) + BatchNorm2d (batch statistics) + ReLU [+ residual]"""

    def __init__(self, ops: _Ops, conv, bn, x: Act, y: Act, res: Optional[Act], name: str):
        ws = ops.ws
        self.ops, self.conv, self.bn, self.x, self.y, self.res, self.name = ops, conv, bn, x, y, res, name
        self.y_s16_only = False       # set by _DoubleConv: y is read by split-fp16 convolutions only -> written as S16, never as fp32
        self.y_s16_too = False        # set by _Stream / TrainEngine: y has fp32 AND S16 readers -> the apply pass writes both
        self.x_is_s16 = False         # ... and its consumer finds the twin of x ready
        self.pool_out = None          # set by _Stream: (S16 twin of the pooled tensor, window positions) - a MaxPool2d(2) follows y
        self._bwd_stats = {}          # id(producer unit) -> (rows, partial rows, segment rows): see `dgrad_stats_for`
        self.cout, self.cin = conv.weight.shape[0], conv.weight.shape[1]
        self.cin_p = _cin_pad(self.cin)
        assert x.c == self.cin_p or x.c == self.cin, (name, x.c, self.cin_p)
        self.kpad = _kpad(9 * self.cin_p)
        self.wp = ws.buf(self.cout, self.kpad)                       # forward filter, packed
        self.craw = ws.act(x.B, x.H, x.W, self.cout)                 # raw conv output (saved for backward)
        self.mean, self.invstd = ws.buf(self.cout), ws.buf(self.cout)
        self.scale, self.shift = ws.buf(self.cout), ws.buf(self.cout)
        self.nblk = ops.lib.ammc_chan_reduce_blocks(x.B * x.H * x.W)
        self.partial = ws.buf(self.nblk, 4, self.cout)               # Q = 2 sums (+ 2 maxima in the fused S16 backward)
        # forward statistics from the convolution's own epilogue where its kernel has one (FUSE_BN_STATS)
        self.stat_rows = (ops.conv_s16_stats_rows(x, self.wp, self.craw, cin=self.cin_p, n=self.cout)
                          if ops.s16 and self.cin_p >= 8 and FUSE_BN_STATS else 0)
        self.stat_partial = ws.buf(self.stat_rows, 2, self.cout) if self.stat_rows else None
        # thousands of rows (8192 at batch 32, 256x256) are first combined in runs of STAT_SEG by many workgroups
        self.stat_seg = ws.buf((self.stat_rows + STAT_SEG - 1) // STAT_SEG, 2, self.cout) if self.stat_rows > 1024 else None
        # backward
        self.dc = ws.act(x.B, x.H, x.W, self.cout)
        self.dwp = ws.zbuf(self.cout, self.kpad)
        self.amax = ws.zbuf(256, dtype=torch.int32)                  # max |dc| slots of this unit (see _Ops.to_s16)
        self.rows = max(64, (self.cin + 63) // 64 * 64) if self.cin >= 32 else 0   # dgrad filter rows
        self.wdp = ws.buf(self.rows, _kpad(9 * self.cout)) if self.rows else None
        self.w16 = self.wd16 = None
        if ops.s16 and self.cin_p >= 8 and PACK_BATCH:
            # S16 images of the forward / input-gradient filters, written by the step's batched pack (TrainEngine)
            self.w16 = ws.buf(self.cout, self.kpad)
            ops.pack_items["fwd"].append((conv.weight, self.w16, self.cout, self.cin, self.cin_p, self.kpad, 0, self.cout))
            if self.rows:
                self.wd16 = ws.buf(self.rows, _kpad(9 * self.cout))
                ops.pack_items["bwd"].append((conv.weight, self.wd16, self.cout, self.cin, self.cout, _kpad(9 * self.cout), 1,
                                              self.rows))

    def forward(self):
        for _ in self.forward_gen():
            raise RuntimeError("synchronised statistics need the lockstep driver (TrainEngine.forward)")

    def forward_gen(self):
        o, lib, s = self.ops, self.ops.lib, self.ops.s
        w = self.conv.weight.detach()
        batched = self.w16 is not None and o.packed["fwd"]
        if not batched:
            _chk(lib.ammc_pack_conv_weight_f32(_ptr(w), self.cout, self.cin, 3, self.cin_p, _ptr(self.wp), s), "pack")
        if o.s16 and self.cin_p >= 8:
            o.conv_s16(self.x, self.wp, self.craw, ntaps=9, cin=self.cin_p, n=self.cout, what=self.name,
                       pre=(o.shadow(self.x), None) if self.x_is_s16 else None, w16=self.w16 if batched else None,
                       stats=self.stat_partial)
        else:
            o.conv(self.x, self.wp, self.craw, ntaps=9, cin=self.cin_p, n=self.cout, what=self.name)
        c = self.craw
        bn = self.bn
        part, nblk, count = self.partial, self.nblk, float(c.B * c.H * c.W)
        if self.stat_rows:
            part, nblk = self.stat_partial, self.stat_rows          # written by the convolution
            if self.stat_seg is not None:
                _chk(lib.ammc_reduce_partials_seg_f32(_ptr(part), nblk, 2 * self.cout, STAT_SEG, 2 * self.cout, _ptr(self.stat_seg), s),
                     "reduce_seg")
                part, nblk = self.stat_seg, self.stat_seg.shape[0]
        else:
            _chk(lib.ammc_bn_stats_f32(c.pix0(), *c.strides, c.B, c.H, c.W, self.cout, _ptr(self.partial), s), "bn_stats")
        world = o.sync_world
        if o.sync_on:
            # synchronised statistics: [2C] sums of every rank are added, the finalizer sees the global batch
            tot = torch.empty(2 * self.cout, device=o.dev, dtype=torch.float32)
            _chk(lib.ammc_reduce_partials_f32(_ptr(part), nblk, 2 * self.cout, 1.0, _ptr(tot), s), "reduce")
            yield tot                                           # summed over the ranks by `_lockstep`
            part, nblk, count = tot, 1, count * world
        _chk(lib.ammc_bn_finalize_f32(_ptr(part), nblk, self.cout, count,
                                      _ptr(bn.weight.detach()), _ptr(bn.bias.detach()), float(bn.eps),
                                      float(bn.momentum if bn.momentum is not None else BN_MOMENTUM),
                                      _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(self.mean),
                                      _ptr(self.invstd), _ptr(self.scale), _ptr(self.shift), s), "bn_finalize")
        if o.nbt is None:
            bn.num_batches_tracked += 1
        else:
            o.nbt.append(bn.num_batches_tracked)              # one fused increment at the end of the forward (TrainEngine)
        r = self.res
        if self.pool_out is not None and r is None and (self.y_s16_only or self.y_s16_too):
            p16, idx = self.pool_out
            _chk(lib.ammc_scale_shift_act_s16_pool_f32(c.pix0(), *c.strides, _ptr(self.scale), _ptr(self.shift),
                                                       None if self.y_s16_only else self.y.pix0(), o.shadow(self.y).pix0(),
                                                       *self.y.strides, p16.pix0(), *p16.strides, idx.data_ptr(), 1, c.B, c.H,
                                                       c.W, self.cout, s), "bn_apply_s16_pool")
            return
        if self.y_s16_only or self.y_s16_too:
            _chk(lib.ammc_scale_shift_act_s16_f32(c.pix0(), *c.strides, _ptr(self.scale), _ptr(self.shift),
                                                  r.pix0() if r is not None else None,
                                                  *(r.strides if r is not None else (0, 0, 0)),
                                                  None if self.y_s16_only else self.y.pix0(),
                                                  o.shadow(self.y).pix0(), *self.y.strides, 1, c.B, c.H, c.W, self.cout, s),
                 "bn_apply_s16")
            return
        _chk(lib.ammc_scale_shift_act_f32(c.pix0(), *c.strides, _ptr(self.scale), _ptr(self.shift),
                                          r.pix0() if r is not None else None, *(r.strides if r is not None else (0, 0, 0)),
                                          self.y.pix0(), *self.y.strides, 1, c.B, c.H, c.W, self.cout, s), "bn_apply")

    def backward(self, dy: Act, da: Optional[Act], da_res: Optional[Act], grads: Dict):
        for _ in self.backward_gen(dy, da, da_res, grads):
            raise RuntimeError("synchronised statistics need the lockstep driver (TrainEngine.backward)")

    def unpool_fusable(self, dy: Act) -> bool:
        """can this unit's backward take its output gradient as `dy + max-pool backward` formed on the fly (`backward_gen`'s
        `unpool`)?  The one-rank split-fp16 path with the row form of the apply kernel; the caller materialises otherwise."""
        o = self.ops
        s16_wgrad = o.s16 and self.cin_p >= 8 and WGRAD_S16
        return bool(FUSE_UNPOOL_BN and FUSE_BN_BWD and o.s16 and s16_wgrad and not o.sync_on and
                    o.lib.ammc_bn_bwd_unpool_supported(self.cout, self.craw.ps, dy.ps, self.dc.ps, self.craw.W))

    def dgrad_stats_for(self, producer: "_ConvBN", dy: Act):
        """(rows, partial, segments) when `producer`'s input-gradient convolution - the kernel that writes this unit's output
        gradient dy - can also leave this unit's BatchNorm-backward partial rows (`AmmcConvDesc.bn_c`), else None.  The
        one-rank split-fp16 path, a halo-patch dgrad kernel with that epilogue, no padded filter rows."""
        o = self.ops
        key = id(producer)
        if key not in self._bwd_stats:
            ok = (FUSE_BN_BWD_STATS and FUSE_BN_BWD and o.s16 and self.cin_p >= 8 and WGRAD_S16 and producer.rows == self.cout
                  and producer.wdp is not None)
            rows = o.conv_s16_stats_rows(producer.dc, producer.wdp, dy, cin=producer.cout, n=producer.rows, bn=self) if ok else 0
            if rows:
                part = o.ws.buf(rows, 4, self.cout)
                seg = o.ws.buf((rows + STAT_SEG - 1) // STAT_SEG, 4, self.cout) if rows > 1024 else None
                self._bwd_stats[key] = (rows, part, seg)
            else:
                self._bwd_stats[key] = None
        st = self._bwd_stats[key]
        return st if st is not None and not o.sync_on else None

    def backward_gen(self, dy: Act, da: Optional[Act], da_res: Optional[Act], grads: Dict, unpool=None, consumer=None,
                     have_stats=None):
        """dy: gradient w.r.t. this unit's output.  Writes da = dgrad (+ da_res) if asked;
        stores the parameter gradients in `grads`.  `unpool` = (dpo, idx): the output was max-pooled in the forward and
        its gradient is dy + MaxPool2d-backward(dpo) by the recorded window positions idx (only if `unpool_fusable`).
        `consumer`: the unit da is the output gradient of - its BatchNorm-backward partial rows come out of this unit's
        dgrad where `consumer.dgrad_stats_for(self, da)` says so; `have_stats` = that tuple on the consumer's own call."""
        o, lib, s = self.ops, self.ops.lib, self.ops.s
        c = self.craw
        world = o.sync_world
        s16_wgrad = o.s16 and self.cin_p >= 8 and WGRAD_S16
        fused_amax = o.s16 and (da is not None or s16_wgrad)
        if unpool is not None:
            assert self.unpool_fusable(dy)
            dpo, idx = unpool
            up = (dpo.pix0(), *dpo.strides, idx.data_ptr(), dpo.H, dpo.W)
            _chk(lib.ammc_bn_bwd_reduce_bound_unpool_f32(c.pix0(), *c.strides, dy.pix0(), *dy.strides, *up, _ptr(self.mean),
                                                         _ptr(self.invstd), _ptr(self.scale), _ptr(self.shift), 1, c.B, c.H,
                                                         c.W, self.cout, _ptr(self.partial), s), "bn_bwd_reduce_bound(unpool)")
            sums = torch.empty(2 * self.cout, device=o.dev, dtype=torch.float32)
            _chk(lib.ammc_bn_bwd_finalize_f32(_ptr(self.partial), self.nblk, self.cout, c.B * c.H * c.W, _ptr(self.scale),
                                              _ptr(sums), self.amax.data_ptr(), s), "bn_bwd_finalize")
            grads[self.bn.bias] = sums[:self.cout]
            grads[self.bn.weight] = sums[self.cout:]
            dc16 = o.shadow(self.dc)
            inv = torch.empty(1024, device=o.dev, dtype=torch.float32)
            _chk(lib.ammc_bn_bwd_apply_s16_unpool_f32(c.pix0(), *c.strides, dy.pix0(), *dy.strides, *up, _ptr(self.mean),
                                                      _ptr(self.invstd), _ptr(self.scale), _ptr(self.shift), _ptr(sums), 1,
                                                      dc16.pix0(), None, *self.dc.strides, c.B, c.H, c.W, self.cout,
                                                      self.amax.data_ptr(), _ptr(inv), 1024, s), "bn_bwd_apply_s16(unpool)")
            pre = (dc16, inv)
        elif fused_amax and not o.sync_on and FUSE_BN_BWD:
            # one rank, S16 consumers: the reduction also bounds max |dc|, so the apply pass writes the S16 twin of dc
            # directly (fp32 dc only where the fp32 weight-gradient kernel still reads it)
            part, nblk = self.partial, self.nblk
            if have_stats is not None:                  # the rows came out of the convolution that wrote dy
                nblk, part, seg = have_stats
                if seg is not None:
                    _chk(lib.ammc_reduce_partials_seg_f32(_ptr(part), nblk, 4 * self.cout, STAT_SEG, 2 * self.cout, _ptr(seg), s),
                         "reduce_seg")
                    part, nblk = seg, seg.shape[0]
            else:
                _chk(lib.ammc_bn_bwd_reduce_bound_f32(c.pix0(), *c.strides, dy.pix0(), *dy.strides, _ptr(self.mean),
                                                      _ptr(self.invstd), _ptr(self.scale), _ptr(self.shift), 1, c.B, c.H, c.W,
                                                      self.cout, _ptr(self.partial), s), "bn_bwd_reduce_bound")
            sums = torch.empty(2 * self.cout, device=o.dev, dtype=torch.float32)
            _chk(lib.ammc_bn_bwd_finalize_f32(_ptr(part), nblk, self.cout, c.B * c.H * c.W, _ptr(self.scale),
                                              _ptr(sums), self.amax.data_ptr(), s), "bn_bwd_finalize")
            grads[self.bn.bias] = sums[:self.cout]
            grads[self.bn.weight] = sums[self.cout:]
            dc16 = o.shadow(self.dc)
            inv = torch.empty(1024, device=o.dev, dtype=torch.float32)
            _chk(lib.ammc_bn_bwd_apply_s16_f32(c.pix0(), *c.strides, dy.pix0(), *dy.strides, _ptr(self.mean),
                                               _ptr(self.invstd), _ptr(self.scale), _ptr(self.shift), _ptr(sums), 1,
                                               dc16.pix0(), None if s16_wgrad else self.dc.pix0(), *self.dc.strides,
                                               c.B, c.H, c.W, self.cout, self.amax.data_ptr(), _ptr(inv), 1024, s),
                 "bn_bwd_apply_s16")
            pre = (dc16, inv)
        else:
            _chk(lib.ammc_bn_bwd_reduce_f32(c.pix0(), *c.strides, dy.pix0(), *dy.strides, _ptr(self.mean), _ptr(self.invstd),
                                            _ptr(self.scale), _ptr(self.shift), 1, c.B, c.H, c.W, self.cout,
                                            _ptr(self.partial), s), "bn_bwd_reduce")
            sums = torch.empty(2 * self.cout, device=o.dev, dtype=torch.float32)
            _chk(lib.ammc_reduce_partials_f32(_ptr(self.partial), self.nblk, 2 * self.cout, 1.0, _ptr(sums), s), "reduce")
            grads[self.bn.bias] = sums[:self.cout]
            grads[self.bn.weight] = sums[self.cout:]
            if o.sync_on:
                # the input gradient needs the sums over the GLOBAL batch; the kernel divides by the local pixel
                # count, so hand it global_sums / world (equal batch per rank).  dgamma / dbeta stay local: the
                # gradient all-reduce averages them like every other parameter.
                g = sums.clone()
                yield g
                sums = g.mul_(1.0 / world)
            # (bn_bwd_apply leaves max |dc| in this unit's slots for the S16 re-encoding)
            _chk(lib.ammc_bn_bwd_apply_f32(c.pix0(), *c.strides, dy.pix0(), *dy.strides, _ptr(self.mean), _ptr(self.invstd),
                                           _ptr(self.scale), _ptr(self.shift), _ptr(sums), 1, self.dc.pix0(),
                                           *self.dc.strides, c.B, c.H, c.W, self.cout,
                                           self.amax.data_ptr() if fused_amax else None, s), "bn_bwd_apply")
            pre = o.to_s16(self.dc, rescale=True, have_amax=True, amax=self.amax) if fused_amax else None   # shared by wgrad and dgrad
        dw = torch.empty_like(self.conv.weight)
        summed = False
        if pre is not None and self.cin_p >= 8 and WGRAD_S16:
            summed = o.wgrad_s16(pre[0], o.shadow(self.x), self.dwp, pre[1], n=self.cout, cin=self.cin_p,
                                 what=self.name + ".wgrad", true_nc=self.cout * self.cin, out_oihw=dw)   # shadow(x): the twin the forward conv left behind
        else:
            o.wgrad(self.dc, self.x, self.dwp, n=self.cout, cin=self.cin_p, ntaps=9, what=self.name + ".wgrad")
        if not summed:
            _chk(lib.ammc_unpack_conv_wgrad_f32(_ptr(self.dwp), self.cout, self.cin, 3, self.cin_p, _ptr(dw), s), "unpack")
        grads[self.conv.weight] = dw
        if da is not None:
            w = self.conv.weight.detach()
            batched = self.wd16 is not None and o.packed["bwd"]
            if not batched:
                _chk(lib.ammc_pack_conv_dgrad_weight_f32(_ptr(w), self.cout, self.cin, self.cout, self.rows,
                                                         _ptr(self.wdp), s), "pack_dgrad")
            if o.s16:
                st = consumer.dgrad_stats_for(self, da) if (consumer is not None and da_res is None and pre is not None) else None
                o.conv_s16(self.dc, self.wdp, da, ntaps=9, cin=self.cout, n=self.rows, res=da_res,
                           what=self.name + ".dgrad", rescale=True, pre=pre, w16=self.wd16 if batched else None,
                           stats=st[1] if st else None, bn=consumer if st else None)
            else:
                o.conv(self.dc, self.wdp, da, ntaps=9, cin=self.cout, n=self.rows, res=da_res, what=self.name + ".dgrad")


class _DoubleConv:
    def __init__(self, ops: _Ops, dc, x: Act, y: Act, res: Optional[Act], name: str):
        seq = dc.conv
        self.mid = ops.ws.act(x.B, x.H, x.W, seq[0].weight.shape[0])
        self.u0 = _ConvBN(ops, seq[0], seq[1], x, self.mid, None, name + ".conv0")
        self.u1 = _ConvBN(ops, seq[3], seq[4], self.mid, y, res, name + ".conv1")
        if ops.s16 and WGRAD_S16 and MID_S16 and self.u1.cin_p >= 8:
            # `mid` is read by conv1 (forward) and by conv1's weight gradient only, both on the S16 kernels
            self.u0.y_s16_only = self.u1.x_is_s16 = True
        self.dmid = ops.ws.act(x.B, x.H, x.W, seq[0].weight.shape[0])

    def emit_twin(self) -> None:
        """the block's output gets its S16 twin from the block's own apply pass"""
        self.u1.y_s16_too = True

    def input_has_twin(self) -> None:
        """the S16 twin of the block's input is complete when the block runs (its producers wrote it)"""
        self.u0.x_is_s16 = True

    def forward_gen(self):
        yield from self.u0.forward_gen()
        yield from self.u1.forward_gen()

    def backward_gen(self, dy: Act, da: Optional[Act], da_res: Optional[Act], grads, unpool=None):
        yield from self.u1.backward_gen(dy, self.dmid, None, grads, unpool=unpool, consumer=self.u0)
        # (conv0's BatchNorm-backward reduction came out of conv1's input-gradient kernel where that kernel has the epilogue)
        yield from self.u0.backward_gen(self.dmid, da, da_res, grads, have_stats=self.u0.dgrad_stats_for(self.u1, self.dmid))


class _Stream:
    """one U-Net stream (`UNet` / `UNetMem_v7`) in training mode"""

    def __init__(self, ops: _Ops, net, B, H, W, has_vq: bool):
        if H < 8 or W < 8:
            raise ValueError(f"frame size {H}x{W}: three 2x2 poolings need at least 8x8")
        ws, lib = ops.ws, ops.lib
        self.ops, self.net, self.B, self.H, self.W, self.has_vq = ops, net, B, H, W, has_vq
        self.cin = net.inc.conv.conv[0].weight.shape[1]
        self.cout = net.outc.weight.shape[0]
        self.x_in = ws.act(B, H, W, _cin_pad(self.cin))
        self.cat = [ws.act(B, H >> i, W >> i, 2 * CHANS[i]) for i in range(3)]
        self.skip = [self.cat[i].slice(0, CHANS[i]) for i in range(3)]
        self.x4 = ws.act(B, H >> 3, W >> 3, 512)
        self.inc = _DoubleConv(ops, net.inc.conv, self.x_in, self.skip[0], None, "inc")
        self.pooled, self.down = [], []
        for i, d in enumerate((net.down1, net.down2, net.down3)):
            p = ws.act(B, H >> (i + 1), W >> (i + 1), CHANS[i])
            out = self.skip[i + 1] if i < 2 else self.x4
            self.pooled.append(p)
            self.down.append(_DoubleConv(ops, d.mpconv[1], p, out, None, f"down{i + 1}"))
        h, w = H >> 3, W >> 3
        self.bottom = self.x4
        if has_vq:
            q = net.vq_down3.quan
            self.q = q
            self.d, self.m, self.k = q.quantize.dim, q.quantize.n_embed, q.quantize.k
            n = B * h * w
            self.n = n
            _check_embed_dim(self.d)
            self.z = ws.act(B, h, w, self.d, halo=0)
            self.qk = ws.act(B, h, w, self.k * self.d, halo=0)
            self.q_one = ws.buf(B, h, w, self.d)
            self.idx = ws.buf(n, self.k, dtype=torch.int32)
            self.nblk_q = lib.ammc_memory_topk_blocks(n)
            self.diff_part = ws.buf(self.nblk_q)
            self.e_md, self.enorm = ws.buf(self.m, self.d), ws.buf(self.m)
            self.enc_wp = ws.buf(self.d, 512)
            self.dec_wp = ws.buf(512, _kpad(self.k * self.d))
            self.x4q = ws.act(B, h, w, 512)
            self.bottom = self.x4q
            self.dz = ws.act(B, h, w, self.d, halo=0)
            self.enc_dwp = ws.zbuf(max(self.d, 32), 512)
            self.dec_dwp = ws.zbuf(512, _kpad(self.k * self.d))
            self.enc_wT = ws.buf(512, _kpad(self.d))
            self.dx4 = ws.act(B, h, w, 512)
        # decoder (inputs are set by the owner: `bottom` may be replaced by the bridge output)
        self.up_mods = (net.up1, net.up2, net.up3)
        self.up_dc: List[_DoubleConv] = []
        self.up_out: List[Act] = []
        self.up_wp, self.up_b4, self.up_dwp, self.up_wT, self.up_amax = [], [], [], [], []
        for j, lvl in enumerate((2, 1, 0)):
            c = CHANS[lvl]
            out = ws.act(B, H >> lvl, W >> lvl, c)
            self.up_out.append(out)
            self.up_dc.append(_DoubleConv(ops, self.up_mods[j].conv, self.cat[lvl], out, None, f"up{j + 1}"))
            self.up_wp.append(ws.buf(4 * c, 2 * c))
            self.up_b4.append(ws.buf(4 * c))
            self.up_dwp.append(ws.zbuf(2 * c, 4 * c))
            self.up_amax.append(ws.zbuf(256, dtype=torch.int32))
            self.up_wT.append(ws.buf(2 * c, 4 * c))
        self.u3 = self.up_out[2]
        self.outc_wp = ws.buf(32, 576)
        self.outc_b = ws.buf(32)
        self.w32 = ws.buf(32, 64, 3, 3)
        # backward buffers
        self.dpre = ws.act(B, H, W, 32)                    # gradient at the output layer: 3 / 2 real channels of 32
        self.du = [ws.act(B, H >> lvl, W >> lvl, CHANS[lvl]) for lvl in (2, 1, 0)]      # grads of up outputs
        self.dcat = [ws.act(B, H >> i, W >> i, 2 * CHANS[i]) for i in range(3)]
        self.dbottom = ws.act(B, h, w, 512)
        self.dskip_tot = [ws.act(B, H >> i, W >> i, CHANS[i]) for i in range(3)]
        self.dpooled = [ws.act(B, H >> (i + 1), W >> (i + 1), CHANS[i]) for i in range(3)]
        # output layer: its gradient buffer is 64 channels wide (3 or 2 real ones) so that the S16 kernels' 64-filter
        # forms apply to its weight- and input-gradient; the fp32 path uses the first 32
        self.outc_dwp = ws.zbuf(64, 576)
        self.outc_amax = ws.zbuf(256, dtype=torch.int32)
        self.outc_wdp = ws.buf(64, _kpad(9 * 64))
        self.scratch = ws.buf(lib.ammc_chan_reduce_blocks(B * H * W) * 512 + 1024)
        # S16 twins written by their producers (TWIN_S16): skips and decoder outputs by the BatchNorm apply pass, pooled
        # tensors by the S16 max-pool, the up half of the concat buffers by the ConvTranspose on conv_gemm_s16
        self.twins = bool(ops.s16 and WGRAD_S16 and TWIN_S16)
        self.pool_idx = None          # per level: window positions of the pooled maxima (a byte each; twins)
        self.pool_fused = [False] * 3  # per level: pooled twin + positions come out of the skip block's apply pass
        self.bottom_twin = False                    # the decoder input's twin is written by ITS producer (set by the owner)
        if self.twins:
            for blk in (self.inc, self.down[0], self.down[1]):
                blk.emit_twin()                     # skip[0..2]: read by the pool and by the decoder's first conv
                # ... as twins; the max-pool backward finds its arg-max on the twin as well (the values the forward's
                # pool compared), so the fp32 skip tensor has no reader and is not written
                blk.u1.y_s16_only = True
            for blk in self.down + self.up_dc:
                blk.input_has_twin()                # pooled[i] / cat[lvl]
            for blk in self.up_dc:
                blk.emit_twin()                     # read by the next ConvTranspose / the output layer
                if CONVT_GRADS_S16:
                    # ... and by nothing else: the next ConvTranspose's forward AND weight gradient, the output layer and
                    # its weight gradient all read the twin - the fp32 tensor would be written for nobody
                    blk.u1.y_s16_only = True
            if not has_vq:
                self.down[2].emit_twin()            # x4 is the decoder's input
                self.bottom_twin = True
            if POOL_IDX:
                self.pool_idx = [torch.empty(q.B, q.H, q.W, q.c, dtype=torch.uint8, device=ops.dev) for q in self.pooled]
                for i, blk in enumerate((self.inc, self.down[0], self.down[1])):
                    sk, c8 = self.skip[i], CHANS[i] >> 3
                    # (the kernel's own verdict - row form available, strides in range, AMMC_ROW_KERNELS - not a copy of its
                    # conditions: a geometry it refuses keeps the two separate passes instead of raising in the forward)
                    p16 = ops.shadow(self.pooled[i])
                    if FUSE_POOL_APPLY and lib.ammc_scale_shift_act_s16_pool_supported(
                            CHANS[i], sk.H, sk.W, blk.u1.craw.rs, blk.u1.craw.ps, sk.rs, sk.ps, p16.ps):
                        blk.u1.pool_out = (p16, self.pool_idx[i])
                        self.pool_fused[i] = True

    # ---- forward pieces -------------------------------------------------------------
    def encode_gen(self, x: torch.Tensor):
        o, lib, s = self.ops, self.ops.lib, self.ops.s
        _chk(lib.ammc_nchw_to_nhwc_f32(_ptr(x), self.B, self.cin, self.H, self.W, self.x_in.pix0(),
                                       *self.x_in.strides, self.x_in.c, s), "nchw_to_nhwc")
        yield from self.inc.forward_gen()
        for i in range(3):
            p, sk = self.pooled[i], self.skip[i]
            if self.pool_fused[i]:
                pass                                # written by inc / down[i - 1]'s apply pass
            elif self.twins:                        # twin -> twin: the fp32 pooled tensor has no reader left
                p16, sk16 = o.shadow(p), o.shadow(sk)
                if POOL_IDX:
                    _chk(lib.ammc_maxpool2x2_s16_idx(sk16.pix0(), *sk16.strides, p16.pix0(), *p16.strides,
                                                     self.pool_idx[i].data_ptr(), p.B, p.H, p.W, p.c, s), "pool_s16")
                else:
                    _chk(lib.ammc_maxpool2x2_s16(sk16.pix0(), *sk16.strides, p16.pix0(), *p16.strides, p.B, p.H, p.W, p.c, s), "pool_s16")
            else:
                _chk(lib.ammc_maxpool2x2_f32(sk.pix0(), *sk.strides, p.pix0(), *p.strides, p.B, p.H, p.W, p.c, s), "pool")
            yield from self.down[i].forward_gen()

    def memory_gen(self):
        """leaves (diff, q_one) in self.mem_out"""
        o, lib, s, q = self.ops, self.ops.lib, self.ops.s, self.q
        qz = q.quantize
        _chk(lib.ammc_pack_conv_weight_f32(_ptr(q.enc.weight.detach()), self.d, 512, 1, 512, _ptr(self.enc_wp), s), "pack")
        _chk(lib.ammc_pack_conv_weight_f32(_ptr(q.dec.weight.detach()), 512, self.k * self.d, 1, _kpad(self.k * self.d),
                                           _ptr(self.dec_wp), s), "pack")
        _chk(lib.ammc_pack_codebook_f32(_ptr(qz.embed), self.d, self.m, _ptr(self.e_md), _ptr(self.enorm), s), "codebook")
        o.conv(self.x4, self.enc_wp, self.z, ntaps=1, cin=512, n=self.d, shift=q.enc.bias.detach(), what="vq.enc")
        _chk(lib.ammc_memory_topk_fwd_f32(_ptr(self.z.buf), _ptr(qz.embed), _ptr(self.e_md), _ptr(self.enorm), self.n,
                                          self.d, self.m, self.k, self.idx.data_ptr(), _ptr(self.qk.buf),
                                          _ptr(self.q_one), _ptr(self.diff_part), s), "memory_topk")
        diff = torch.empty(1, device=o.dev, dtype=torch.float32)
        _chk(lib.ammc_sum_partials_f32(_ptr(self.diff_part), self.nblk_q, 1.0 / float(self.n * self.d), _ptr(diff), s),
             "diff")
        # EMA update AFTER the lookups (they use the pre-update codebook, unet.py:291-309); e_md keeps the
        # pre-update rows, which is what the backward's commit gradient needs
        if o.sync_on:
            counts = torch.empty(self.m, device=o.dev, dtype=torch.float32)
            sums = torch.empty((self.d, self.m), device=o.dev, dtype=torch.float32)
            _chk(lib.ammc_codebook_count_f32(_ptr(self.z.buf), self.idx.data_ptr(), self.k, self.n, self.d, self.m,
                                             _ptr(counts), _ptr(sums), s), "codebook_count")
            yield [counts, sums]
            _chk(lib.ammc_codebook_ema_apply_f32(_ptr(counts), _ptr(sums), self.d, self.m, float(qz.decay),
                                                 float(1 - qz.decay), float(qz.eps), _ptr(qz.cluster_size),
                                                 _ptr(qz.embed_avg), _ptr(qz.embed), s), "codebook_ema_apply")
        else:
            _chk(lib.ammc_codebook_ema_f32(_ptr(self.z.buf), self.idx.data_ptr(), self.k, self.n, self.d, self.m,
                                           float(qz.decay), float(1 - qz.decay), float(qz.eps), _ptr(qz.cluster_size),
                                           _ptr(qz.embed_avg), _ptr(qz.embed), s), "codebook_ema")
        o.conv(self.qk, self.dec_wp, self.x4q, ntaps=1, cin=self.k * self.d, n=512, shift=q.dec.bias.detach(),
               res=self.x4, what="vq.dec")
        self.mem_out = (diff, self.q_one.clone())

    def decode_gen(self, bottom: Act):
        """leaves the predicted frame in self.out"""
        o, lib, s = self.ops, self.ops.lib, self.ops.s
        self.dec_in = bottom
        y = bottom
        for j, lvl in enumerate((2, 1, 0)):
            c = CHANS[lvl]
            m = self.up_mods[j]
            _chk(lib.ammc_pack_convt_weight_f32(_ptr(m.up.weight.detach()), 2 * c, c, _ptr(self.up_wp[j]), s), "pack")
            self.up_b4[j].copy_(m.up.bias.detach().repeat(4))
            dst = self.cat[lvl].slice(c, c)
            if self.twins:
                # S16 in (the producer's twin), S16 out (the up half of the concat buffer's twin: its only readers are the
                # first conv of the block and that conv's weight gradient, both on the split-fp16 kernels)
                has = self.bottom_twin if j == 0 else True
                o.conv_s16(y, self.up_wp[j], o.shadow(dst), ntaps=1, cin=2 * c, n=4 * c, shift=self.up_b4[j], up=2, cgroup=c,
                           what=f"up{j + 1}.up", pre=(o.shadow(y), None) if has else None, y_s16=True)
            else:
                (o.conv_s16 if o.s16 and CONVT_S16 else o.conv)(y, self.up_wp[j], dst, ntaps=1, cin=2 * c,
                                                              n=4 * c, shift=self.up_b4[j], up=2, cgroup=c, what=f"up{j + 1}.up")
            yield from self.up_dc[j].forward_gen()
            y = self.up_out[j]
        net = self.net
        self.w32[:self.cout].copy_(net.outc.weight.detach())          # rows >= cout stay zero from allocation
        _chk(lib.ammc_pack_conv_weight_f32(_ptr(self.w32), 32, 64, 3, 64, _ptr(self.outc_wp), s), "pack")
        self.outc_b[:self.cout].copy_(net.outc.bias.detach())
        out = torch.empty((self.B, self.cout, self.H, self.W), device=o.dev, dtype=torch.float32)
        d = AmmcConvDesc()
        u3 = self.u3
        d.shift, d.y = _ptr(self.outc_b), _ptr(out)
        d.batch, d.height, d.width = self.B, self.H, self.W
        d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act, d.n_store = 64, 9, 32, 1, 32, ACT_TANH, self.cout
        d.y_bs, d.y_rs, d.y_ps, d.y_cs = self.cout * self.H * self.W, self.W, 1, self.H * self.W
        if o.s16:
            u16 = o.shadow(u3) if self.twins else o.to_s16(u3)[0]       # also the A operand of the weight gradient later
            w16 = torch.empty_like(self.outc_wp)
            _chk(lib.ammc_split_rows_f32(_ptr(self.outc_wp), self.outc_wp.numel(), _ptr(w16), s), "split_rows(w)")
            d.x, d.w, d.y_f32 = u16.tap0(), _ptr(w16), 1
            d.x_bs, d.x_rs, d.x_ps = u16.strides
            _chk(lib.ammc_conv_gemm_s16(C.byref(d), s), "outc")
        else:
            d.x, d.w = u3.tap0(), _ptr(self.outc_wp)
            d.x_bs, d.x_rs, d.x_ps = u3.strides
            _chk(lib.ammc_conv_gemm_f32(C.byref(d), s), "outc")
        self.out = out

    # ---- backward pieces ------------------------------------------------------------
    def decode_backward_gen(self, dout: torch.Tensor, grads):
        """from d(tanh output) down to the gradient of the decoder's bottom input (self.dbottom); fills dcat[*]"""
        o, lib, s, net = self.ops, self.ops.lib, self.ops.s, self.net
        dout = dout.contiguous()
        dp = self.dpre
        _chk(lib.ammc_tanh_bwd_nhwc_f32(_ptr(dout), _ptr(self.out), self.B, self.cout, self.H, self.W, dp.pix0(),
                                        *dp.strides, 32, s), "tanh_bwd")
        grads[net.outc.bias] = o.chan_sum(dp, 32, self.scratch)[:self.cout]
        dw = torch.empty_like(net.outc.weight)
        summed = False
        if o.s16 and WGRAD_S16:
            pre = o.to_s16(dp, rescale=True, amax=self.outc_amax)
            summed = o.wgrad_s16(pre[0], o.shadow(self.u3), self.outc_dwp, pre[1], n=32, cin=64, what="outc.wgrad",
                                 true_nc=self.cout * 64, out_oihw=dw)
        else:
            pre = None
            o.wgrad(dp.slice(0, 32), self.u3, self.outc_dwp, n=32, cin=64, ntaps=9, what="outc.wgrad")
        if not summed:
            _chk(lib.ammc_unpack_conv_wgrad_f32(_ptr(self.outc_dwp), self.cout, 64, 3, 64, _ptr(dw), s), "unpack")
        grads[net.outc.weight] = dw
        _chk(lib.ammc_pack_conv_dgrad_weight_f32(_ptr(net.outc.weight.detach()), self.cout, 64, 32, 64,
                                                 _ptr(self.outc_wdp), s), "pack_dgrad")
        if o.s16:
            o.conv_s16(dp, self.outc_wdp, self.du[2], ntaps=9, cin=32, n=64, what="outc.dgrad", rescale=True, pre=pre)
        else:
            o.conv(dp, self.outc_wdp, self.du[2], ntaps=9, cin=32, n=64, what="outc.dgrad")
        for j in (2, 1, 0):
            lvl = (2, 1, 0)[j]
            c = CHANS[lvl]
            m = self.up_mods[j]
            yield from self.up_dc[j].backward_gen(self.du[j], self.dcat[lvl], None, grads)
            x_in = self.dec_in if j == 0 else self.up_out[j - 1]
            # gradient of the ConvTranspose output: the top-left 2h x 2w of the skip-sized tensor (`up.forward` pads an
            # odd level on the right / bottom, models/unet_parts.py; the pad's gradient is dropped)
            dys = self.dcat[lvl].slice(c, c).crop(2 * x_in.H, 2 * x_in.W)
            both16 = self.twins and CONVT_GRADS_S16 and (self.bottom_twin if j == 0 else True)
            pre = None
            if both16:
                # one pass: bias gradient + max |g| of the slice; one strided re-encoding; both gradient kernels read it
                amax = self.up_amax[j]
                nb = lib.ammc_chan_reduce_blocks(dys.B * dys.H * dys.W)
                _chk(lib.ammc_chan_sum_absmax_f32(dys.pix0(), *dys.strides, dys.B, dys.H, dys.W, c, _ptr(self.scratch),
                                                  amax.data_ptr(), s), "chan_sum_absmax")
                bsum = torch.empty(c, device=o.dev, dtype=torch.float32)
                _chk(lib.ammc_reduce_partials_f32(_ptr(self.scratch), nb, c, 1.0, _ptr(bsum), s), "reduce_partials")
                grads[m.up.bias] = bsum
                dys16 = o.shadow(dys)
                inv = torch.empty(1024, device=o.dev, dtype=torch.float32)
                _chk(lib.ammc_split_scaled_strided_f32(dys.pix0(), *dys.strides, dys16.pix0(), *dys16.strides, dys.B, dys.H,
                                                       dys.W, c, amax.data_ptr(), _ptr(inv), 1024, s), "split_scaled_strided")
                pre = (dys16, inv)
                o.wgrad_s16(o.shadow(x_in), dys16, self.up_dwp[j], inv, n=2 * c, cin=c, ntaps=4, a_step=2,
                            what=f"up{j + 1}.up.wgrad")
            else:
                grads[m.up.bias] = o.chan_sum(dys, c, self.scratch)
                o.wgrad(x_in, dys, self.up_dwp[j], n=2 * c, cin=c, ntaps=4, a_step=2, what=f"up{j + 1}.up.wgrad")
            dwt = torch.empty_like(m.up.weight)
            _chk(lib.ammc_unpack_convt_wgrad_f32(_ptr(self.up_dwp[j]), 2 * c, c, _ptr(dwt), s), "unpack_convt")
            grads[m.up.weight] = dwt
            _chk(lib.ammc_transpose_pad_f32(_ptr(self.up_wp[j]), 4 * c, 2 * c, 4 * c, _ptr(self.up_wT[j]), s), "transpose")
            dst = self.dbottom if j == 0 else self.du[j - 1]
            if both16:
                o.conv_s16(dys, self.up_wT[j], dst, ntaps=4, cin=c, n=2 * c, x_step=2, what=f"up{j + 1}.up.dgrad", pre=pre)
            elif o.s16 and CONVT_S16:
                o.conv_s16(dys, self.up_wT[j], dst, ntaps=4, cin=c, n=2 * c, x_step=2, rescale=True,
                           what=f"up{j + 1}.up.dgrad")
            else:
                o.conv(dys, self.up_wT[j], dst, ntaps=4, cin=c, n=2 * c, x_step=2, what=f"up{j + 1}.up.dgrad")

    def memory_backward(self, dq4: Act, ddiff: Optional[torch.Tensor], dq_one: Optional[torch.Tensor], grads) -> Act:
        """gradient through dec / commit term / enc / residual; returns d(x4)"""
        o, lib, s, q = self.ops, self.ops.lib, self.ops.s, self.q
        kd = self.k * self.d
        grads[q.dec.bias] = o.chan_sum(dq4, 512, self.scratch)
        o.wgrad(dq4, self.qk, self.dec_dwp, n=512, cin=kd, ntaps=1, what="vq.dec.wgrad")
        dw = torch.empty_like(q.dec.weight)
        _chk(lib.ammc_unpack_conv_wgrad_f32(_ptr(self.dec_dwp), 512, kd, 1, _kpad(kd), _ptr(dw), s), "unpack")
        grads[q.dec.weight] = dw
        dq = dq_one.contiguous() if dq_one is not None else None
        _chk(lib.ammc_commit_bwd_f32(_ptr(self.z.buf), _ptr(self.e_md), self.idx.data_ptr(), self.k,
                                     _ptr(ddiff) if ddiff is not None else None, _ptr(dq) if dq is not None else None,
                                     _ptr(self.dz.buf), self.n, self.d, s), "commit_bwd")
        grads[q.enc.bias] = o.chan_sum(self.dz, self.d, self.scratch)
        o.wgrad(self.dz, self.x4, self.enc_dwp, n=max(self.d, 32), cin=512, ntaps=1, what="vq.enc.wgrad")
        dwe = torch.empty_like(q.enc.weight)
        _chk(lib.ammc_unpack_conv_wgrad_f32(_ptr(self.enc_dwp), self.d, 512, 1, 512, _ptr(dwe), s), "unpack")
        grads[q.enc.weight] = dwe
        _chk(lib.ammc_transpose_pad_f32(_ptr(self.enc_wp), self.d, 512, _kpad(self.d), _ptr(self.enc_wT), s), "transpose")
        o.conv(self.dz, self.enc_wT, self.dx4, ntaps=1, cin=_kpad(self.d), n=512, res=dq4, what="vq.enc.dgrad")
        return self.dx4

    def encode_backward_gen(self, dx4: Act, grads):
        o, lib, s = self.ops, self.ops.lib, self.ops.s
        dy, unpool = dx4, None
        blocks = (self.inc, self.down[0], self.down[1])
        for i in (2, 1, 0):
            yield from self.down[i].backward_gen(dy, self.dpooled[i], None, grads, unpool=unpool)
            sk, dpo, add, out = self.skip[i], self.dpooled[i], self.dcat[i].slice(0, CHANS[i]), self.dskip_tot[i]
            unpool = None
            if self.twins and POOL_IDX and blocks[i].u1.unpool_fusable(add):
                # the gradient of skip[i] (skip path + max-pool backward) is formed inside the BatchNorm-backward passes of
                # the block that produced skip[i]: no pass writes it, none reads it back
                dy, unpool = add, (dpo, self.pool_idx[i])
                continue
            if self.twins and POOL_IDX:
                _chk(lib.ammc_maxpool2x2_bwd_idx_f32(self.pool_idx[i].data_ptr(), dpo.pix0(), *dpo.strides, add.pix0(),
                                                     *add.strides, out.pix0(), *out.strides, dpo.B, dpo.H, dpo.W, sk.H, sk.W,
                                                     dpo.c, s), "maxpool_bwd(idx)")
            elif self.twins:
                sk16 = o.shadow(sk)
                _chk(lib.ammc_maxpool2x2_bwd_s16x_f32(sk16.pix0(), *sk16.strides, dpo.pix0(), *dpo.strides, add.pix0(),
                                                      *add.strides, out.pix0(), *out.strides, dpo.B, dpo.H, dpo.W, sk.H, sk.W,
                                                      dpo.c, s), "maxpool_bwd(s16)")
            else:
                _chk(lib.ammc_maxpool2x2_bwd_f32(sk.pix0(), *sk.strides, dpo.pix0(), *dpo.strides, add.pix0(), *add.strides,
                                                 out.pix0(), *out.strides, dpo.B, dpo.H, dpo.W, sk.H, sk.W, dpo.c, s), "maxpool_bwd")
            dy = out
        yield from self.inc.backward_gen(dy, None, None, grads, unpool=unpool)


class TrainEngine:
    """training-mode forward/backward of `UNet`, `UNetMem_v7` or `twostream`"""

    def __init__(self, module, kind: str, precision: Optional[str] = None):
        precision = precision or TRAIN_PRECISION
        if precision not in ("fp32", "s16"):
            raise ValueError("train precision must be 'fp32' (exact fp32 MFMA) or 's16' (split-fp16 MFMA 3x3 layers)")
        self.module, self.kind, self.precision = module, kind, precision
        self._ws: Dict = {}
        self.generation = 0

    def _get(self, B, H, W, device):
        key = (B, H, W, device)
        st = self._ws.get(key)
        if st is None:
            ws = _WS(device)
            ops = _Ops(ws, self.precision)
            m = self.module
            if self.kind == "twostream":
                r = _Stream(ops, m.rgb, B, H, W, True)
                o = _Stream(ops, m.op, B, H, W, True)
                h, w = H >> 3, W >> 3
                xb, yb = ws.act(B, h, w, 512), ws.act(B, h, w, 512)
                o2f = _DoubleConv(ops, m.bridge.O2F, o.x4q, xb, r.x4q, "bridge.O2F")
                f2o = _DoubleConv(ops, m.bridge.F20, r.x4q, yb, o.x4q, "bridge.F20")
                if r.twins:
                    o2f.emit_twin()                 # xb / yb feed the first ConvTranspose of their stream
                    f2o.emit_twin()
                    r.bottom_twin = o.bottom_twin = True
                    if CONVT_GRADS_S16:             # (forward and weight gradient of that ConvTranspose: twin readers both)
                        o2f.u1.y_s16_only = f2o.u1.y_s16_only = True
                st = dict(ops=ops, streams=[r, o], o2f=o2f, f2o=f2o, xb=xb, yb=yb,
                          dzx=ws.act(B, h, w, 512), dzy=ws.act(B, h, w, 512))
            else:
                s = _Stream(ops, m, B, H, W, self.kind == "unetmem")
                st = dict(ops=ops, streams=[s])
            st["bytes"] = ws.bytes
            self._ws[key] = st
        return st

    def forward(self, *inputs: torch.Tensor):
        x0 = inputs[0]
        if not x0.is_cuda:
            raise _lib.AmmcHipError("the HIP path needs CUDA/HIP tensors; there is no CPU fallback")
        B, _, H, W = x0.shape
        st = self._get(B, H, W, x0.device)
        sync = getattr(self.module, "_sync_stats", None)          # parallel.sync_statistics(model, group)
        st["ops"].sync_group = sync[1] if sync and sync[0] else False
        st["ops"].sync_force = bool(sync and sync[0] and len(sync) > 2 and sync[2])
        self.generation += 1
        st["generation"] = self.generation
        xs = [x.detach().float().contiguous() for x in inputs]
        streams: List[_Stream] = st["streams"]
        diffs, qs, outs = [], [], []
        # same order as the reference's forward (unet.py:981-1007): it fixes the order of the
        # in-place buffer updates
        # (with synchronised statistics the two streams advance layer by layer and share each collective, `_lockstep`;
        # their buffers are disjoint, so the order between the streams does not matter - within a stream it is kept)
        ops = st["ops"]
        ops.nbt = []
        ops.packed["fwd"] = ops.packed["bwd"] = False
        if ops.s16 and PACK_BATCH:
            ops.run_pack("fwd")                    # every 3x3 forward filter of the step -> S16, one launch
        def both(*gens, grads=None):
            if not _side_by_side(ops, grads, *gens):
                _lockstep(ops, *gens)

        both(*[s.encode_gen(x) for s, x in zip(streams, xs)])
        vq = [s for s in streams if s.has_vq]
        _lockstep(ops, *[s.memory_gen() for s in vq])
        for s in vq:
            diffs.append(s.mem_out[0])
            qs.append(s.mem_out[1])
        if self.kind == "twostream":
            both(st["o2f"].forward_gen(), st["f2o"].forward_gen())
            bottoms = [st["xb"], st["yb"]]
        else:
            bottoms = [streams[0].bottom]
        both(*[s.decode_gen(b) for s, b in zip(streams, bottoms)])
        if ops.nbt:
            torch._foreach_add_(ops.nbt, 1)
        ops.nbt = None
        outs = [s.out for s in streams]
        self._last = st
        if hasattr(self.module, "_param_epoch"):
            self.module._param_epoch += 1          # buffers changed through raw pointers: invalidate eval packs
        if self.kind == "unet":
            return (outs[0],)
        if self.kind == "unetmem":
            return outs[0], diffs[0], qs[0]
        return outs[0], outs[1], diffs[0], diffs[1], qs[0], qs[1]

    def backward(self, generation: int, gouts) -> Dict:
        st = self._last
        if st.get("generation") != generation:
            raise RuntimeError("HIP training path: backward() called for a forward whose workspace has been reused "
                               "by a later forward of the same shape (one forward per backward is supported)")
        streams: List[_Stream] = st["streams"]
        grads: Dict = {}
        st["ops"].ws.zero_step()            # weight-gradient accumulators and max-|g| slots: one memset per 64-MB chunk
        if st["ops"].s16 and PACK_BATCH:
            st["ops"].run_pack("bwd")           # every input-gradient filter -> S16, one launch
        reducer = getattr(self.module, "_grad_reducer", None)     # parallel.BucketedGradReducer or None
        sent = set()

        def g(t):
            return t if t is not None else None

        def stage_done():
            """hand the gradients finished since the last call to the all-reduce (overlaps the rest)"""
            if reducer is not None:
                new = [v for k, v in grads.items() if id(k) not in sent]
                sent.update(id(k) for k in grads)
                reducer.push(new)

        if self.kind == "twostream":
            d_rgb, d_op, dd_r, dd_o, dq_r, dq_o = gouts
            r, o = streams
            ops = st["ops"]
            douts = [dout if dout is not None else torch.zeros_like(s.out) for s, dout in ((r, d_rgb), (o, d_op))]
            solo = reducer is None          # one rank, no all-reduce to feed: the two streams side by side (`_side_by_side`)
            if solo and _side_by_side(ops, grads, r.decode_backward_gen(douts[0], grads), o.decode_backward_gen(douts[1], grads)):
                pass
            elif ops.sync_on:
                _lockstep(ops, r.decode_backward_gen(douts[0], grads), o.decode_backward_gen(douts[1], grads))
                stage_done()
            else:                                   # one stream after the other: its gradients leave for the all-reduce early
                for s, dout in zip((r, o), douts):
                    _lockstep(ops, s.decode_backward_gen(dout, grads))
                    stage_done()
            # x = zx + O2F(zy); y = zy + F20(zx): dzy = dyb + dgrad_O2F(dxb), dzx = dxb + dgrad_F20(dyb)
            bridge = (st["o2f"].backward_gen(r.dbottom, st["dzy"], o.dbottom, grads),
                      st["f2o"].backward_gen(o.dbottom, st["dzx"], r.dbottom, grads))
            if not (solo and _side_by_side(ops, grads, *bridge)):
                _lockstep(ops, *bridge)
            stage_done()
            dx4r = r.memory_backward(st["dzx"], g(dd_r), g(dq_r), grads)
            dx4o = o.memory_backward(st["dzy"], g(dd_o), g(dq_o), grads)
            if solo and _side_by_side(ops, grads, r.encode_backward_gen(dx4r, grads), o.encode_backward_gen(dx4o, grads)):
                pass
            elif ops.sync_on:
                _lockstep(ops, r.encode_backward_gen(dx4r, grads), o.encode_backward_gen(dx4o, grads))
            else:
                _lockstep(ops, r.encode_backward_gen(dx4r, grads))
                stage_done()
                _lockstep(ops, o.encode_backward_gen(dx4o, grads))
        else:
            s = streams[0]
            dout = gouts[0] if gouts[0] is not None else torch.zeros_like(s.out)
            _lockstep(st["ops"], s.decode_backward_gen(dout, grads))
            db = s.dbottom
            if self.kind == "unetmem":
                db = s.memory_backward(db, g(gouts[1]), g(gouts[2]), grads)
            _lockstep(st["ops"], s.encode_backward_gen(db, grads))
        stage_done()
        if reducer is not None:
            reducer.finish()
        return {k.data_ptr(): v for k, v in grads.items()}


class BlockEngine:
    """Training-mode forward / backward of ONE building block called on its own, as the reference's sub-modules can be
    (`double_conv`, `inconv`, `down`, `up`, `bridge`: models/unet.py:8-59, 956-965) - the same units the whole-model
    engine is made of, with the module-boundary layout changes around them and, unlike the whole-model engine, the
    gradients of the block's INPUTS (a stand-alone block sits inside somebody else's autograd graph).  Input gradients
    need >= 32 input channels (the input-gradient convolution's tiles); for thinner inputs - the 12 / 6-channel clips
    of `inconv` - the input gets no gradient, as data does not need one."""

    def __init__(self, module, kind: str, precision: Optional[str] = None):
        precision = precision or TRAIN_PRECISION
        if kind not in ("double_conv", "down", "up", "bridge"):
            raise ValueError(kind)
        self.module, self.kind, self.precision = module, kind, precision
        self._ws: Dict = {}
        self.generation = 0

    def _get(self, shapes, device):
        key = (shapes, device)
        st = self._ws.get(key)
        if st is not None:
            return st
        ws = _WS(device)
        ops = _Ops(ws, self.precision)
        m, kind = self.module, self.kind
        B, C, H, W = shapes[0]
        st = dict(ops=ops)
        if kind in ("double_conv", "down"):
            dc = m if kind == "double_conv" else m.mpconv[1]
            cin = dc.conv[0].weight.shape[1]
            if C != cin:
                raise ValueError(f"{kind}: {C} input channels, the block has {cin}")
            st["x"] = ws.act(B, H, W, _cin_pad(cin))
            src = st["x"]
            if kind == "down":
                st["pooled"] = src = ws.act(B, H // 2, W // 2, _cin_pad(cin))
                st["dpooled"] = ws.act(B, H // 2, W // 2, _cin_pad(cin))
            cout = dc.conv[0].weight.shape[0]
            st["y"] = ws.act(B, src.H, src.W, cout)
            st["dy"] = ws.act(B, src.H, src.W, cout)
            st["dc"] = _DoubleConv(ops, dc, src, st["y"], None, kind)
            st["dx"] = ws.act(B, H, W, _cin_pad(cin)) if cin >= 32 else None
        elif kind == "up":
            (_, c2, h, w), (_, c, H2, W2) = shapes
            if c2 != 2 * c or (H2 // 2, W2 // 2) != (h, w):
                raise NotImplementedError("up in training mode: x2 must have half the channels and twice the size of x1 "
                                          "(+1 for an odd level, padded on the right / bottom as `up.forward` does)")
            st["x1"] = ws.act(B, h, w, c2)
            st["cat"] = ws.act(B, H2, W2, c2)
            st["y"], st["dy"] = ws.act(B, H2, W2, m.conv.conv[0].weight.shape[0]), ws.act(B, H2, W2, m.conv.conv[0].weight.shape[0])
            st["dc"] = _DoubleConv(ops, m.conv, st["cat"], st["y"], None, "up")
            st["dcat"], st["dx1"] = ws.act(B, H2, W2, c2), ws.act(B, h, w, c2)
            st["wp"], st["b4"] = ws.buf(4 * c, c2), ws.buf(4 * c)
            st["dwp"], st["wT"] = ws.zbuf(c2, 4 * c), ws.buf(c2, 4 * c)
            st["scratch"] = ws.buf(ops.lib.ammc_chan_reduce_blocks(B * H2 * W2) * 512 + 1024)
        else:                                   # bridge: x = zx + O2F(zy), y = zy + F20(zx)
            st["zx"], st["zy"] = ws.act(B, H, W, C), ws.act(B, H, W, C)
            st["xb"], st["yb"] = ws.act(B, H, W, C), ws.act(B, H, W, C)
            st["dxb"], st["dyb"] = ws.act(B, H, W, C), ws.act(B, H, W, C)
            st["dzx"], st["dzy"] = ws.act(B, H, W, C), ws.act(B, H, W, C)
            st["o2f"] = _DoubleConv(ops, m.O2F, st["zy"], st["xb"], st["zx"], "bridge.O2F")
            st["f2o"] = _DoubleConv(ops, m.F20, st["zx"], st["yb"], st["zy"], "bridge.F20")
        self._ws[key] = st
        return st

    @staticmethod
    def _load(ops, x: torch.Tensor, a: Act):
        x = x.detach().float().contiguous()
        _chk(ops.lib.ammc_nchw_to_nhwc_f32(_ptr(x), a.B, x.shape[1], a.H, a.W, a.pix0(), *a.strides, a.c, ops.s), "nchw_to_nhwc")

    @staticmethod
    def _store(ops, a: Act, c: Optional[int] = None) -> torch.Tensor:
        c = c or a.c
        y = torch.empty((a.B, c, a.H, a.W), device=a.buf.device, dtype=torch.float32)
        _chk(ops.lib.ammc_nhwc_to_nchw_f32(a.pix0(), *a.strides, a.B, c, a.H, a.W, _ptr(y), ops.s), "nhwc_to_nchw")
        return y

    def forward(self, *inputs: torch.Tensor):
        if not inputs[0].is_cuda:
            raise _lib.AmmcHipError("the HIP path needs CUDA/HIP tensors; there is no CPU fallback")
        st = self._get(tuple(tuple(x.shape) for x in inputs), inputs[0].device)
        ops, lib = st["ops"], st["ops"].lib
        ops.sync_group = False
        self.generation += 1
        st["generation"] = self.generation
        self._last = st
        if self.kind in ("double_conv", "down"):
            self._load(ops, inputs[0], st["x"])
            if self.kind == "down":
                x, p = st["x"], st["pooled"]
                _chk(lib.ammc_maxpool2x2_f32(x.pix0(), *x.strides, p.pix0(), *p.strides, p.B, p.H, p.W, p.c, ops.s), "pool")
            _lockstep(ops, st["dc"].forward_gen())
            return (self._store(ops, st["y"]),)
        if self.kind == "up":
            m, c = self.module, inputs[1].shape[1]
            self._load(ops, inputs[0], st["x1"])
            self._load(ops, inputs[1], st["cat"].slice(0, c))
            _chk(lib.ammc_pack_convt_weight_f32(_ptr(m.up.weight.detach()), 2 * c, c, _ptr(st["wp"]), ops.s), "pack")
            st["b4"].copy_(m.up.bias.detach().repeat(4))
            ops.conv(st["x1"], st["wp"], st["cat"].slice(c, c), ntaps=1, cin=2 * c, n=4 * c, shift=st["b4"], up=2, cgroup=c,
                     what="up.up")
            _lockstep(ops, st["dc"].forward_gen())
            return (self._store(ops, st["y"]),)
        self._load(ops, inputs[0], st["zx"])
        self._load(ops, inputs[1], st["zy"])
        _lockstep(ops, st["o2f"].forward_gen(), st["f2o"].forward_gen())
        return self._store(ops, st["xb"]), self._store(ops, st["yb"])

    def backward(self, generation: int, gouts):
        """-> ([gradient per input or None], {parameter data_ptr: gradient})"""
        st = self._last
        if st.get("generation") != generation:
            raise RuntimeError("HIP training path: backward() called for a forward whose workspace has been reused")
        ops, lib = st["ops"], st["ops"].lib
        ops.ws.zero_step()
        grads: Dict = {}
        if self.kind in ("double_conv", "down"):
            self._load(ops, gouts[0], st["dy"])
            if self.kind == "double_conv":
                _lockstep(ops, st["dc"].backward_gen(st["dy"], st["dx"], None, grads))
                dx = self._store(ops, st["dx"], st["dc"].u0.cin) if st["dx"] is not None else None
            else:
                dpo = st["dpooled"] if st["dx"] is not None else None
                _lockstep(ops, st["dc"].backward_gen(st["dy"], dpo, None, grads))
                dx = None
                if dpo is not None:
                    x, out = st["x"], st["dx"]
                    _chk(lib.ammc_maxpool2x2_bwd_f32(x.pix0(), *x.strides, dpo.pix0(), *dpo.strides, None, 0, 0, 0,
                                                     out.pix0(), *out.strides, dpo.B, dpo.H, dpo.W, x.H, x.W, dpo.c, ops.s), "maxpool_bwd")
                    dx = self._store(ops, out, st["dc"].u0.cin)
            ins = [dx]
        elif self.kind == "up":
            m = self.module
            c = m.up.weight.shape[1]
            self._load(ops, gouts[0], st["dy"])
            _lockstep(ops, st["dc"].backward_gen(st["dy"], st["dcat"], None, grads))
            dx2 = self._store(ops, st["dcat"].slice(0, c), c)
            dys = st["dcat"].slice(c, c).crop(2 * st["x1"].H, 2 * st["x1"].W)
            grads[m.up.bias] = ops.chan_sum(dys, c, st["scratch"])
            ops.wgrad(st["x1"], dys, st["dwp"], n=2 * c, cin=c, ntaps=4, a_step=2, what="up.up.wgrad")
            dwt = torch.empty_like(m.up.weight)
            _chk(lib.ammc_unpack_convt_wgrad_f32(_ptr(st["dwp"]), 2 * c, c, _ptr(dwt), ops.s), "unpack_convt")
            grads[m.up.weight] = dwt
            _chk(lib.ammc_transpose_pad_f32(_ptr(st["wp"]), 4 * c, 2 * c, 4 * c, _ptr(st["wT"]), ops.s), "transpose")
            ops.conv(dys, st["wT"], st["dx1"], ntaps=4, cin=c, n=2 * c, x_step=2, what="up.up.dgrad")
            ins = [self._store(ops, st["dx1"]), dx2]
        else:
            dxb = gouts[0] if gouts[0] is not None else torch.zeros_like(gouts[1])
            dyb = gouts[1] if gouts[1] is not None else torch.zeros_like(gouts[0])
            self._load(ops, dxb, st["dxb"])
            self._load(ops, dyb, st["dyb"])
            # x = zx + O2F(zy); y = zy + F20(zx): dzy = dyb + dgrad_O2F(dxb), dzx = dxb + dgrad_F20(dyb)
            _lockstep(ops, st["o2f"].backward_gen(st["dxb"], st["dzy"], st["dyb"], grads),
                      st["f2o"].backward_gen(st["dyb"], st["dzx"], st["dxb"], grads))
            ins = [self._store(ops, st["dzx"]), self._store(ops, st["dzy"])]
        return ins, {k.data_ptr(): v for k, v in grads.items()}


class MemoryBlockEngine:
    """Training-mode forward / backward of the memory block called on its own, as the reference's modules can be:
    `Quantize_topk` (models/unet.py:282-316: the lookups, the commit term, the EMA update of the codebook buffers
    whenever `self.training`), `enc_quan_dec_topk` (:318-331) and `enc_quan_dec_res_topk` (:379-387).  One autograd node:
    forward = [1x1 enc] -> fused distance / top-k / gather kernel -> EMA update AFTER the lookups -> [1x1 dec (+ x)];
    backward = the gradient of the commit term 2 (z - e_idx) / (n d) plus the straight-through gradient of `quantize`
    (`ammc_commit_bwd_f32`), the weight / bias / input gradients of the two 1x1 convolutions.  The gathered rows
    (`quantize_topk`) come out of a buffer: no gradient reaches anything through them, exactly as in the reference."""

    def __init__(self, module, kind: str, precision: Optional[str] = None):
        if kind not in ("quantize", "vq", "vq_res"):
            raise ValueError(kind)
        self.module, self.kind = module, kind
        self.precision = "fp32"                     # 1x1 convolutions and the lookups run on the exact-fp32 kernels
        self._ws: Dict = {}
        self.generation = 0

    def _get(self, shape, device):
        key = (shape, device)
        st = self._ws.get(key)
        if st is not None:
            return st
        ws = _WS(device, zchunk=1 << 18)            # 1 MB slabs: two 1x1 accumulators
        ops = _Ops(ws, "fp32")
        lib = ops.lib
        qz = self.module if self.kind == "quantize" else self.module.quantize
        d, m, k = qz.dim, qz.n_embed, qz.k
        if self.kind != "quantize":
            _check_embed_dim(d)
        st = dict(ops=ops, d=d, m=m, k=k)
        if self.kind == "quantize":
            B, h, w, dd = shape
            if dd != d:
                raise ValueError(f"Quantize_topk: last dimension {dd}, the codebook has {d}")
        else:
            B, C, h, w = shape
            cin = self.module.enc.weight.shape[1]
            if C != cin or C % 32:
                raise NotImplementedError(f"memory block in training mode: {C} input channels (the block has {cin}; multiples of 32)")
            st["x"], st["y"], st["dy"], st["dx"] = ws.act(B, h, w, C), ws.act(B, h, w, C), ws.act(B, h, w, C), ws.act(B, h, w, C)
            st["enc_wp"], st["dec_wp"] = ws.buf(d, C), ws.buf(C, _kpad(k * d))
            st["enc_dwp"], st["dec_dwp"] = ws.zbuf(max(d, 32), C), ws.zbuf(C, _kpad(k * d))
            st["enc_wT"] = ws.buf(C, _kpad(d))
            st["scratch"] = ws.buf(lib.ammc_chan_reduce_blocks(B * h * w) * 512 + 1024)
        n = B * h * w
        st.update(B=B, h=h, w=w, n=n, z=ws.act(B, h, w, d, halo=0), qk=ws.act(B, h, w, k * d, halo=0), dz=ws.act(B, h, w, d, halo=0),
                  q_one=ws.buf(B, h, w, d), idx=ws.buf(n, k, dtype=torch.int32), nblk=lib.ammc_memory_topk_blocks(n),
                  e_md=ws.buf(m, d), enorm=ws.buf(m))
        st["part"] = ws.buf(st["nblk"])
        self._ws[key] = st
        return st

    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise _lib.AmmcHipError("the HIP path needs CUDA/HIP tensors; there is no CPU fallback")
        st = self._get(tuple(x.shape), x.device)
        o, lib = st["ops"], st["ops"].lib
        o.sync_group = False
        s = o.s
        self.generation += 1
        st["generation"] = self.generation
        self._last = st
        d, m, k, n = st["d"], st["m"], st["k"], st["n"]
        if self.kind == "quantize":
            qz = self.module
            st["z"].buf.view(-1).copy_(x.detach().float().reshape(-1))
        else:
            q, qz = self.module, self.module.quantize
            C = st["x"].c
            BlockEngine._load(o, x, st["x"])
            _chk(lib.ammc_pack_conv_weight_f32(_ptr(q.enc.weight.detach()), d, C, 1, C, _ptr(st["enc_wp"]), s), "pack")
            _chk(lib.ammc_pack_conv_weight_f32(_ptr(q.dec.weight.detach()), C, k * d, 1, _kpad(k * d), _ptr(st["dec_wp"]), s), "pack")
            o.conv(st["x"], st["enc_wp"], st["z"], ntaps=1, cin=C, n=d, shift=q.enc.bias.detach(), what="vq.enc")
        _chk(lib.ammc_pack_codebook_f32(_ptr(qz.embed), d, m, _ptr(st["e_md"]), _ptr(st["enorm"]), s), "codebook")
        _chk(lib.ammc_memory_topk_fwd_f32(_ptr(st["z"].buf), _ptr(qz.embed), _ptr(st["e_md"]), _ptr(st["enorm"]), n, d, m, k,
                                          st["idx"].data_ptr(), _ptr(st["qk"].buf), _ptr(st["q_one"]), _ptr(st["part"]), s), "memory_topk")
        diff = torch.empty(1, device=o.dev, dtype=torch.float32)
        _chk(lib.ammc_sum_partials_f32(_ptr(st["part"]), st["nblk"], 1.0 / float(n * d), _ptr(diff), s), "diff")
        # the EMA update AFTER the lookups (unet.py:291-309); e_md keeps the pre-update rows for the backward
        _chk(lib.ammc_codebook_ema_f32(_ptr(st["z"].buf), st["idx"].data_ptr(), k, n, d, m, float(qz.decay), float(1 - qz.decay),
                                       float(qz.eps), _ptr(qz.cluster_size), _ptr(qz.embed_avg), _ptr(qz.embed), s), "codebook_ema")
        if self.kind == "quantize":
            return st["qk"].buf.view(st["B"], st["h"], st["w"], k * d).clone(), diff[0], st["q_one"].clone()
        o.conv(st["qk"], st["dec_wp"], st["y"], ntaps=1, cin=k * d, n=st["x"].c, shift=q.dec.bias.detach(),
               res=st["x"] if self.kind == "vq_res" else None, what="vq.dec")
        return BlockEngine._store(o, st["y"]), diff, st["q_one"].clone()

    def backward(self, generation: int, gouts):
        st = self._last
        if st.get("generation") != generation:
            raise RuntimeError("HIP training path: backward() called for a forward whose workspace has been reused")
        o, lib = st["ops"], st["ops"].lib
        s = o.s
        o.ws.zero_step()
        d, m, k, n = st["d"], st["m"], st["k"], st["n"]
        g_out, g_diff, g_q1 = gouts
        grads: Dict = {}
        ddiff = g_diff.detach().float().reshape(1).contiguous() if g_diff is not None else None
        dq = g_q1.detach().float().contiguous() if g_q1 is not None else None
        _chk(lib.ammc_commit_bwd_f32(_ptr(st["z"].buf), _ptr(st["e_md"]), st["idx"].data_ptr(), k,
                                     _ptr(ddiff) if ddiff is not None else None, _ptr(dq) if dq is not None else None,
                                     _ptr(st["dz"].buf), n, d, s), "commit_bwd")
        if self.kind == "quantize":
            return [st["dz"].buf.view(st["B"], st["h"], st["w"], d).clone()], {}
        q = self.module
        C, kd = st["x"].c, k * d
        if g_out is not None:
            BlockEngine._load(o, g_out, st["dy"])
            grads[q.dec.bias] = o.chan_sum(st["dy"], C, st["scratch"])
            o.wgrad(st["dy"], st["qk"], st["dec_dwp"], n=C, cin=kd, ntaps=1, what="vq.dec.wgrad")
            dw = torch.empty_like(q.dec.weight)
            _chk(lib.ammc_unpack_conv_wgrad_f32(_ptr(st["dec_dwp"]), C, kd, 1, _kpad(kd), _ptr(dw), s), "unpack")
            grads[q.dec.weight] = dw
        grads[q.enc.bias] = o.chan_sum(st["dz"], d, st["scratch"])
        o.wgrad(st["dz"], st["x"], st["enc_dwp"], n=max(d, 32), cin=C, ntaps=1, what="vq.enc.wgrad")
        dwe = torch.empty_like(q.enc.weight)
        _chk(lib.ammc_unpack_conv_wgrad_f32(_ptr(st["enc_dwp"]), d, C, 1, C, _ptr(dwe), s), "unpack")
        grads[q.enc.weight] = dwe
        _chk(lib.ammc_transpose_pad_f32(_ptr(st["enc_wp"]), d, C, _kpad(d), _ptr(st["enc_wT"]), s), "transpose")
        o.conv(st["dz"], st["enc_wT"], st["dx"], ntaps=1, cin=_kpad(d), n=C,
               res=st["dy"] if (self.kind == "vq_res" and g_out is not None) else None, what="vq.enc.dgrad")
        return [BlockEngine._store(o, st["dx"])], {k_.data_ptr(): v for k_, v in grads.items()}


class MemoryBlockFunction(torch.autograd.Function):
    """(engine, input, *params) -> (out | quantize_topk, diff, quantize_one), differentiable w.r.t. input and params"""

    @staticmethod
    def forward(ctx, engine: MemoryBlockEngine, x, *params):
        outs = tuple(o.detach() for o in engine.forward(x))
        ctx.engine, ctx.generation, ctx.params = engine, engine.generation, params
        ctx.set_materialize_grads(False)
        if engine.kind == "quantize":
            ctx.mark_non_differentiable(outs[0])
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        ins, grads = ctx.engine.backward(ctx.generation, gouts)
        out = [None, ins[0] if ctx.needs_input_grad[1] else None]
        for p in ctx.params:
            out.append(grads.get(p.data_ptr()))
        return tuple(out)


class BlockFunction(torch.autograd.Function):
    """(engine, n_inputs, *inputs, *params) -> the block's outputs, differentiable w.r.t. inputs and params"""

    @staticmethod
    def forward(ctx, engine: BlockEngine, n_inputs: int, *tensors):
        outs = tuple(o.detach() for o in engine.forward(*tensors[:n_inputs]))
        ctx.engine, ctx.generation, ctx.params, ctx.n_inputs = engine, engine.generation, tensors[n_inputs:], n_inputs
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *gouts):
        ins, grads = ctx.engine.backward(ctx.generation, gouts)
        out = [None, None] + [g if ctx.needs_input_grad[2 + i] else None for i, g in enumerate(ins)]
        for p in ctx.params:
            out.append(grads.get(p.data_ptr()))
        return tuple(out)


class HipPathFunction(torch.autograd.Function):
    """(engine, n_inputs, *inputs, *params) -> the model's outputs, differentiable w.r.t. params"""

    @staticmethod
    def forward(ctx, engine: TrainEngine, n_inputs: int, *tensors):
        inputs, params = tensors[:n_inputs], tensors[n_inputs:]
        # (fresh tensor objects for autograd to hang its node on: the engine keeps the ones it made - `stream.out`, `mem_out` -
        # and a node on THOSE objects would close a cycle engine -> tensor -> grad_fn -> ctx -> engine through the C++ autograd
        # graph, which Python's collector cannot see: every training forward's ~60-GB workspace then outlived its model
        # (round 6: the GPU suite's live set reached 284 GB))
        outs = tuple(o.detach() for o in engine.forward(*inputs))
        ctx.engine, ctx.generation, ctx.params, ctx.n_inputs = engine, engine.generation, params, n_inputs
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        grads = ctx.engine.backward(ctx.generation, gouts)
        out = [None, None] + [None] * ctx.n_inputs
        for p in ctx.params:
            out.append(grads.get(p.data_ptr()))
        return tuple(out)
