"""Launch plans for the HIP path: which kernel runs on which buffer, in what order.

An engine belongs to one model (`UNet`, `UNetMem_v7` or `twostream`).  It keeps
  * the pre-packed weights (K-major conv filters, folded eval-BatchNorm, the
    slot-major codebook copy) - rebuilt when a parameter/buffer changes;
  * one workspace per input shape: every activation of the network as a
    halo-padded NHWC fp32 buffer in HBM, allocated once (halo zeroed once:
    kernels only ever write interiors);
  * one launch plan per input shape: the fully resolved argument lists of every
    C-ABI call, so a forward is a flat loop of ctypes calls on the current HIP
    stream (and can be captured into a hipGraph, `capture()`).

Layout decisions: the skip tensor of each encoder level and
the ConvTranspose output of the matching decoder level are channel slices of
ONE concat buffer, so torch.cat (reference unet.py:57) costs nothing; max-pool
reads the skip slice in place.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_TANH, AmmcConvDesc

BN_EPS = 1e-5
# AMMC_GRAPH=1: eval forwards replay a captured hipGraph (one host call per forward).  Off by default: measured
# on MI355X the ~100 launches of a forward are NOT launch-bound (batch 1: 1.58 ms eager vs 1.58 ms replayed; the
# chain of dependent small kernels is the floor), and a replay costs an extra copy of the inputs into static buffers.
USE_GRAPH = os.environ.get("AMMC_GRAPH", "0") != "0"


def _ptr(t: torch.Tensor, elem_off: int = 0) -> int:
    return t.data_ptr() + 4 * elem_off


class Act:
    """A [B, H, W, C] activation: a channel slice of an NHWC buffer with `halo` zero pixels."""

    def __init__(self, buf: torch.Tensor, B: int, H: int, W: int, c: int, c_off: int = 0, halo: int = 1):
        self.buf, self.B, self.H, self.W, self.c, self.c_off, self.halo = buf, B, H, W, c, c_off, halo
        self.c_total = buf.shape[-1]
        self.ps = self.c_total
        self.rs = (W + 2 * halo) * self.ps
        self.bs = (H + 2 * halo) * self.rs

    @property
    def strides(self) -> Tuple[int, int, int]:
        return self.bs, self.rs, self.ps

    def tap0(self) -> int:
        """address of window tap (0,0) of pixel (0,0): the halo corner"""
        assert self.halo == 1
        return _ptr(self.buf, self.c_off)

    def pix0(self) -> int:
        """address of pixel (0,0) itself"""
        return _ptr(self.buf, self.halo * (self.rs + self.ps) + self.c_off)

    def slice(self, c_off: int, c: int) -> "Act":
        a = Act(self.buf, self.B, self.H, self.W, c, self.c_off + c_off, self.halo)
        a.rs, a.bs = self.rs, self.bs
        return a

    def crop(self, H: int, W: int) -> "Act":
        """the top-left H x W pixels of the same buffer (strides unchanged)"""
        a = Act(self.buf, self.B, self.H, self.W, self.c, self.c_off, self.halo)
        a.rs, a.bs = self.rs, self.bs           # a crop of a crop keeps the BUFFER's strides
        a.H, a.W = H, W
        return a

    def interior(self) -> torch.Tensor:
        """NHWC view [B,H,W,c] (for tests / debugging)"""
        h = self.halo
        return self.buf[:, h:h + self.H, h:h + self.W, self.c_off:self.c_off + self.c]


def s16_variant(d: AmmcConvDesc) -> str:
    """the kernel `ammc_conv_gemm_s16` launches for this descriptor, by the name rocprofv3 reports
    (`ammc_conv_gemm_s16_variant`: the library's own dispatch code with the launch replaced by the label, so
    bench labels and test coverage cannot drift from what runs).  Needs no GPU."""
    buf = C.create_string_buffer(96)
    _lib.check(_lib.load().ammc_conv_gemm_s16_variant(C.byref(d), buf, 96), "conv_gemm_s16_variant")
    return buf.value.decode()


def _kpad(k: int) -> int:
    return (k + 31) // 32 * 32


def _cin_pad(c: int) -> int:
    p = 4
    while p < c:
        p *= 2
    return p


# eval forward of `twostream`: the rgb and the flow stream's launches on two HIP streams, crossed in front of the bridge
# (the step is energy-bound at ~1350 of 1400 W on average: two queues let one stream's low-power kernels and kernel tails
# run beside the other's convolutions; 6.51 -> 6.38 ms at batch 16)
EVAL_LANES = os.environ.get("AMMC_EVAL_LANES", "1") != "0"


class Plan:
    """A flat list of resolved C-ABI calls."""

    def __init__(self):
        self.calls: List[Tuple] = []
        self.meta: List[dict] = []
        self.keep: List = []
        self.lane = None          # set by the builder around a section: 0 / 1 = the rgb / flow stream's own work, None = joint

    def add(self, fn, *args, name: str = "", flops: float = 0.0, nbytes: float = 0.0, kernel: str = "", dyn=None):
        """`dyn(ctx) -> args`: arguments resolved per forward (calls that read the caller's input tensors directly)"""
        self.calls.append((fn, dyn if dyn is not None else args, name))
        self.meta.append(dict(name=name, flops=flops, bytes=nbytes, kernel=kernel, lane=self.lane))

    def run(self, stream: int):
        for fn, args, name in self.calls:
            rc = fn(*args, stream)
            if rc != 0:
                _lib.check(rc, name or fn.__name__)

    def run_timed(self, stream: int):
        """Same launches, each bracketed by HIP events on the launch stream.
        Returns [(meta, milliseconds)]; used by bench.py for the roofline figures."""
        evs = []
        for fn, args, name in self.calls:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args, stream)
            e1.record()
            if rc != 0:
                _lib.check(rc, name or fn.__name__)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return [(m, e0.elapsed_time(e1)) for m, (e0, e1) in zip(self.meta, evs)]


# the memory block of an S16 plan as one launch (ammc_memory_block_s16); AMMC_FUSED_MEMORY=0: the five-launch chain (A/Bs)
FUSED_MEMORY = os.environ.get("AMMC_FUSED_MEMORY", "1") != "0"


class _Packer:
    """weight pre-packing through the C ABI"""

    def __init__(self, device, s16: bool = False):
        self.lib = _lib.load()
        self.device = device
        self.s16 = s16                  # filters as (hi, lo) half pairs for ammc_conv_gemm_s16
        # range verdicts of the S16 packs (round 6): [0] = a packed filter beyond the half range, [1..] = one per codebook
        # (`codebook_s16`); read ONCE per parameter version by EvalEngine._ensure_packs
        self.flags = torch.zeros(8, device=device, dtype=torch.int32) if s16 else None
        self.nflags = 1

    def _split(self, t: torch.Tensor) -> torch.Tensor:
        """packed filter [N][Kpad] -> S16"""
        if not self.s16:
            return t
        out = torch.empty_like(t)
        _lib.check(self.lib.ammc_split_rows_guarded_f32(_ptr(t), t.numel(), _ptr(out), self.flags.data_ptr(), self.stream()),
                   "split_rows")
        return out

    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def conv(self, w: torch.Tensor, ksize: int) -> Tuple[torch.Tensor, int]:
        cout, cin = w.shape[0], w.shape[1]
        cin_p = _cin_pad(cin) if ksize == 3 else _kpad(cin)
        out = torch.empty((cout, _kpad(ksize * ksize * cin_p)), device=self.device, dtype=torch.float32)
        w = w.detach().contiguous()
        _lib.check(self.lib.ammc_pack_conv_weight_f32(_ptr(w), cout, cin, ksize, cin_p, _ptr(out), self.stream()),
                   "pack_conv_weight")
        return self._split(out), cin_p

    def convt(self, w: torch.Tensor) -> torch.Tensor:
        cin, co = w.shape[0], w.shape[1]
        out = torch.empty((4 * co, cin), device=self.device, dtype=torch.float32)
        w = w.detach().contiguous()
        _lib.check(self.lib.ammc_pack_convt_weight_f32(_ptr(w), cin, co, _ptr(out), self.stream()), "pack_convt")
        return self._split(out)

    def first_conv(self, conv0: torch.nn.Conv2d) -> Optional[torch.Tensor]:
        """filter image of `ammc_conv_first_s16` (csrc/conv_first_s16.hip), or None when the layer is not its case"""
        cout, cin = conv0.weight.shape[0], conv0.weight.shape[1]
        if cout != 64 or cin > 16:
            return None
        img = torch.empty(self.lib.ammc_first_conv_image_floats(), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.ammc_pack_first_conv_f32(_ptr(conv0.weight.detach().contiguous()), cout, cin, _ptr(img),
                                                     self.stream()), "pack_first_conv")
        return img

    def up_conv(self, conv0: torch.nn.Conv2d, convt: torch.nn.ConvTranspose2d, scale: torch.Tensor,
                shift: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """composed filters + border-class shifts of `ammc_conv_up_s16` (csrc/conv_up_s16.hip): the transposed conv of an
        `up` block folded into the up half of the 3x3 conv that follows it"""
        n, c2 = conv0.weight.shape[0], conv0.weight.shape[1]
        c = c2 // 2
        w2 = torch.empty((n, 16 * c2), device=self.device, dtype=torch.float32)
        shift9 = torch.empty((9, n), device=self.device, dtype=torch.float32)
        w3, wt, bt = conv0.weight.detach().contiguous(), convt.weight.detach().contiguous(), convt.bias.detach().contiguous()
        _lib.check(self.lib.ammc_pack_up_conv_f32(_ptr(w3), _ptr(wt), _ptr(bt), _ptr(scale), _ptr(shift), n, c, _ptr(w2),
                                                  _ptr(shift9), self.stream()), "pack_up_conv")
        return self._split(w2), shift9

    def bn(self, bn: torch.nn.BatchNorm2d) -> Tuple[torch.Tensor, torch.Tensor]:
        c = bn.num_features
        scale = torch.empty(c, device=self.device, dtype=torch.float32)
        shift = torch.empty(c, device=self.device, dtype=torch.float32)
        _lib.check(self.lib.ammc_bn_fold_f32(_ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                                             _ptr(bn.running_mean), _ptr(bn.running_var), float(bn.eps), c,
                                             _ptr(scale), _ptr(shift), self.stream()), "bn_fold")
        return scale, shift

    def codebook(self, embed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        d, m = embed.shape
        e_md = torch.empty((m, d), device=self.device, dtype=torch.float32)
        enorm = torch.empty(m, device=self.device, dtype=torch.float32)
        _lib.check(self.lib.ammc_pack_codebook_f32(_ptr(embed), d, m, _ptr(e_md), _ptr(enorm), self.stream()),
                   "pack_codebook")
        return e_md, enorm

    def codebook_s16(self, embed: torch.Tensor) -> Tuple[torch.Tensor, int]:
        """[d/8][mpad][8 hi | 8 lo] halfs: the slot operand of `ammc_memory_topk_fwd_s16`, and the index of its range
        verdict in `self.flags` (raised when a slot does not fit the half range: the reference's from-scratch EMA state
        holds |embed| ~ 1e5 in every slot no row has hit yet, models/unet.py:277-280, 298-309)"""
        d, m = embed.shape
        mpad = (m + 31) // 32 * 32
        out = torch.empty((d // 8, mpad, 16), device=self.device, dtype=torch.float16)
        fi = self.nflags
        self.nflags += 1
        _lib.check(self.lib.ammc_pack_codebook_s16_guarded(_ptr(embed), d, m, out.data_ptr(),
                                                           self.flags.data_ptr() + 4 * fi, self.stream()), "pack_codebook_s16")
        return out, fi


class _DoubleConvPack:
    def __init__(self, pk: _Packer, dc):
        seq = dc.conv
        self.w0, self.cin_p = pk.conv(seq[0].weight, 3)
        self.cin = seq[0].weight.shape[1]
        self.s0, self.b0 = pk.bn(seq[1])
        self.w1, _ = pk.conv(seq[3].weight, 3)
        self.s1, self.b1 = pk.bn(seq[4])
        self.cout = seq[0].weight.shape[0]


class _StreamPack:
    """packed parameters of one U-Net stream (`UNet` / `UNetMem_v7`)"""

    def __init__(self, pk: _Packer, net):
        self.inc = _DoubleConvPack(pk, net.inc.conv)
        self.first = pk.first_conv(net.inc.conv.conv[0]) if pk.s16 else None      # S16 plans: the first layer from NCHW
        self.down = [_DoubleConvPack(pk, d.mpconv[1]) for d in (net.down1, net.down2, net.down3)]
        self.up = []
        self.up_fused = []        # S16 plans: (composed filters, border-class shifts) of ammc_conv_up_s16 per decoder level
        for u in (net.up1, net.up2, net.up3):
            dcp = _DoubleConvPack(pk, u.conv)
            self.up.append((pk.convt(u.up.weight), u.up.bias.detach(), dcp))
            self.up_fused.append(pk.up_conv(u.conv.conv[0], u.up, dcp.s0, dcp.b0) if pk.s16 else None)
        # `outc` rides in a 32-column MFMA tile: pad the 2-3 filters (and the bias) to 32 rows
        self.cout = net.outc.weight.shape[0]
        w32 = torch.zeros((32,) + tuple(net.outc.weight.shape[1:]), device=pk.device, dtype=torch.float32)
        w32[:self.cout].copy_(net.outc.weight.detach())
        self.outc_w, _ = pk.conv(w32, 3)
        self.outc_b = torch.zeros(32, device=pk.device, dtype=torch.float32)
        self.outc_b[:self.cout].copy_(net.outc.bias.detach())
        self.cin = net.inc.conv.conv[0].weight.shape[1]
        self.vq = None
        if hasattr(net, "vq_down3"):
            q = net.vq_down3.quan
            enc_w, _ = pk.conv(q.enc.weight, 1)
            dec_w, _ = pk.conv(q.dec.weight, 1)
            e_md, enorm = pk.codebook(q.quantize.embed)
            e_s16, e_flag = pk.codebook_s16(q.quantize.embed) if (pk.s16 and q.quantize.dim == 64) else (None, None)
            self.vq = dict(enc_w=enc_w, enc_b=q.enc.bias.detach(), dec_w=dec_w, dec_b=q.dec.bias.detach(),
                           embed=q.quantize.embed, e_md=e_md, enorm=enorm, d=q.quantize.dim,
                           m=q.quantize.n_embed, k=q.quantize.k, e_s16=e_s16, e_s16_flag=e_flag)


class _Builder:
    """appends resolved kernel calls to a Plan"""

    def __init__(self, plan: Plan, device, s16: bool = False, arena: Optional[List[torch.Tensor]] = None):
        self.plan = plan
        self.lib = _lib.load()
        self.device = device
        self.s16 = s16
        # buffers of a plan built earlier for the same frame size and a batch at least as large: the build order
        # is deterministic, so allocation i of this plan can be the first B images of allocation i of that one
        # (explicit batch strides everywhere; halos are zero and never written).  A short last batch of a
        # sub-video then costs no new multi-GB workspace.
        self.arena = arena
        self.made: List[torch.Tensor] = []
        self.overflow = torch.zeros(1, device=device, dtype=torch.int32) if s16 else None   # set by S16 epilogues
        # fp32 scratch for split-K on small-M layers (latency at small batch); shared by all layers of a plan
        self.splitk = torch.empty(8 << 20, device=device, dtype=torch.float32) if s16 else None
        self.splitk_b = None      # ... a second one for the flow stream's lane (made when a plan has lanes: they run concurrently)
        self.conv_fn = self.lib.ammc_conv_gemm_s16 if s16 else self.lib.ammc_conv_gemm_f32
        self.kname = "conv_gemm_s16" if s16 else "conv_gemm_f32"

    def buf(self, *shape) -> torch.Tensor:
        # zeros once: kernels write interiors only, so the halo stays zero forever
        i = len(self.made)
        t = None
        if self.arena is not None and i < len(self.arena):
            big = self.arena[i]
            if big.dim() >= 3 and len(shape) == big.dim() and tuple(big.shape[1:]) == tuple(shape[1:]) \
                    and big.shape[0] >= shape[0]:
                t = big[:shape[0]]
        if t is None:
            t = torch.zeros(shape, device=self.device, dtype=torch.float32)
        self.made.append(t)
        self.plan.keep.append(t)
        return t

    def act(self, B, H, W, c, halo=1) -> Act:
        return Act(self.buf(B, H + 2 * halo, W + 2 * halo, c), B, H, W, c, 0, halo)

    def conv(self, x: Act, w: torch.Tensor, y: Act, *, ntaps: int, cin: int, n: int, scale=None, shift=None,
             act=ACT_NONE, res: Optional[Act] = None, up: int = 1, cgroup: Optional[int] = None, name="conv",
             cin_true: Optional[int] = None, y_f32: bool = False, pool: Optional[Act] = None):
        d = AmmcConvDesc()
        d.y_f32 = 1 if (y_f32 and self.s16) else 0
        d.overflow_flag = self.overflow.data_ptr() if self.s16 else None
        if self.s16 and os.environ.get("AMMC_S16_SPLITK", "1") != "0":
            ws = self.splitk
            if self.plan.lane in (1, 3):
                if self.splitk_b is None:
                    self.splitk_b = torch.empty_like(self.splitk)
                    self.plan.keep.append(self.splitk_b)
                ws = self.splitk_b
            d.splitk_ws, d.splitk_ws_floats = ws.data_ptr(), ws.numel()
        d.x = x.tap0() if ntaps == 9 else x.pix0()
        d.w = _ptr(w)
        d.y = y.pix0()
        d.scale = _ptr(scale) if scale is not None else None
        d.shift = _ptr(shift) if shift is not None else None
        d.res = res.pix0() if res is not None else None
        d.batch, d.height, d.width = x.B, x.H, x.W
        d.cin, d.ntaps, d.n, d.up = cin, ntaps, n, up
        d.cgroup = cgroup if cgroup is not None else n
        d.act = act
        d.x_bs, d.x_rs, d.x_ps = x.strides
        d.y_bs, d.y_rs, d.y_ps = y.strides
        if res is not None:
            d.r_bs, d.r_rs, d.r_ps = res.strides
        self.plan.keep.append(d)
        self.plan.keep.extend(t for t in (w, scale, shift) if t is not None)
        # algorithmic work of this launch: 2*M*N*K with the TRUE channel count;
        # bytes = input read once + output written once (+ residual read), weights excluded
        m_pix = x.B * x.H * x.W
        ct = cin_true if cin_true is not None else cin
        flops = 2.0 * m_pix * n * ntaps * ct
        nbytes = 4.0 * m_pix * (ct + n + (n if res is not None else 0))
        if self.s16:
            kernel = s16_variant(d)                     # asked BEFORE a fused-pool output is attached (see double_conv)
        else:
            kernel = "conv_gemm_f32<%s>" % ("128x32" if n == 32 else "128x128" if n % 128 == 0 else "128x64")
        if pool is not None:                 # fused 2x2 max-pool output (the halo-patch kernel only; the caller checked)
            d.pool_y = pool.pix0()
            d.pool_bs, d.pool_rs, d.pool_ps = pool.strides
        self.plan.add(self.conv_fn, C.byref(d), name=name, flops=flops, nbytes=nbytes, kernel=kernel)
        return d

    def outc_desc(self, x: Act, w: torch.Tensor, bias32: torch.Tensor, cout: int) -> AmmcConvDesc:
        """`outc` + tanh with an NCHW store; the output pointer is patched per call (fresh tensor)"""
        d = AmmcConvDesc()
        d.x, d.w, d.shift = x.tap0(), _ptr(w), _ptr(bias32)
        d.batch, d.height, d.width = x.B, x.H, x.W
        d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act, d.n_store = x.c, 9, 32, 1, 32, ACT_TANH, cout
        d.y_f32 = 1 if self.s16 else 0
        d.x_bs, d.x_rs, d.x_ps = x.strides
        d.y_bs, d.y_rs, d.y_ps, d.y_cs = cout * x.H * x.W, x.W, 1, x.H * x.W
        self.plan.keep.extend([d, w, bias32])
        if self.s16:
            d.y = d.x                          # any aligned non-null address: the label query launches nothing
            self.outc_kernel = s16_variant(d)
            d.y = None
        else:
            self.outc_kernel = "conv_gemm_f32<128x32>"
        return d

    def double_conv(self, x: Act, p: _DoubleConvPack, mid: Act, y: Act, res: Optional[Act] = None, name="dc",
                    pool: Optional[Act] = None, skip_first: bool = False) -> bool:
        """`pool`: where the 2x2 max-pool of the output goes; returns True when the second conv stored it itself.
        `skip_first`: the first conv is launched by the caller (`ammc_conv_first_s16`, straight from the NCHW input)"""
        if not skip_first:
            self.conv(x, p.w0, mid, ntaps=9, cin=p.cin_p, n=p.cout, scale=p.s0, shift=p.b0, act=ACT_RELU,
                      name=f"{name}.conv0", cin_true=p.cin)
        fused = False
        if pool is not None and res is None and self.s16 and os.environ.get("AMMC_FUSE_POOL", "1") != "0":
            # the second output exists in the halo-patch kernel only: ask the library whether this layer gets it
            probe = AmmcConvDesc()
            probe.x, probe.w, probe.y = mid.tap0(), _ptr(p.w1), y.pix0()
            probe.batch, probe.height, probe.width = mid.B, mid.H, mid.W
            probe.cin, probe.ntaps, probe.n, probe.up, probe.cgroup = p.cout, 9, p.cout, 1, p.cout
            probe.x_bs, probe.x_rs, probe.x_ps = mid.strides
            probe.y_bs, probe.y_rs, probe.y_ps = y.strides
            fused = s16_variant(probe).startswith("conv_tap_s16")
        self.conv(mid, p.w1, y, ntaps=9, cin=p.cout, n=p.cout, scale=p.s1, shift=p.b1, act=ACT_RELU, res=res,
                  name=f"{name}.conv1", pool=pool if fused else None)
        return fused

    def up_conv_eligible(self, skip: Act, n: int) -> bool:
        """does csrc/conv_up_s16.hip take this decoder level?  (mirrors the checks of `ammc_conv_up_s16`)"""
        mode = os.environ.get("AMMC_UP_FUSED", "1")
        if not (self.s16 and mode != "0" and skip.W % 32 == 0 and skip.H % 8 == 0 and skip.c % 32 == 0
                and (n == 64 or n % 128 == 0)):
            return False
        # one workgroup per 8 x 32 output tile and 128 filters: below ~128 of them the chip is empty and the two
        # launches it replaces (implicit GEMMs with split-K) win - measured at 256x256: batch 1, up1 (32 tiles) 203 us
        # against 74, up2 (64) 105 against 69; batch 2, up2 (128) 110 against 114; up3 (256 per clip) always ahead
        # (AMMC_UP_FUSED=2 forces the fused kernel)
        tiles = skip.B * (skip.H // 8) * (skip.W // 32) * max(1, n // 128)
        return tiles >= 128 or mode == "2"

    def up_conv(self, x2: Act, skip: Act, p: _DoubleConvPack, fused, y: Act, name="up"):
        """ConvTranspose2d(x2) + cat([skip, .]) + conv3x3 + BN + ReLU as ONE launch (reference unet.py:50-59, first conv
        of `up.conv`): the transposed conv is folded into the filters of the up half (`_Packer.up_conv`)"""
        w2, shift9 = fused
        d = AmmcConvDesc()
        d.x, d.w, d.y = skip.tap0(), _ptr(p.w0), y.pix0()
        d.scale = _ptr(p.s0)
        d.batch, d.height, d.width = skip.B, skip.H, skip.W
        d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act = skip.c, 9, p.cout, 1, p.cout, ACT_RELU
        d.x_bs, d.x_rs, d.x_ps = skip.strides
        d.y_bs, d.y_rs, d.y_ps = y.strides
        d.overflow_flag = self.overflow.data_ptr()
        self.plan.keep.extend([d, p.w0, p.s0, w2, shift9])
        m_pix = skip.B * skip.H * skip.W
        # algorithmic work of what this launch REPLACES: the transposed conv + the 3x3 conv over 2c channels
        flops = 2.0 * m_pix * p.cout * 9 * 2 * skip.c + 2.0 * (m_pix // 4) * x2.c * 4 * skip.c
        nbytes = 4.0 * (m_pix * (skip.c + p.cout) + (m_pix // 4) * x2.c)
        self.plan.add(self.lib.ammc_conv_up_s16, C.byref(d), x2.tap0(), *x2.strides, x2.c, _ptr(w2), _ptr(shift9),
                      name=name, flops=flops, nbytes=nbytes, kernel="conv_up_s16<%d>" % (2 if p.cout == 64 else 4))

    def maxpool(self, x: Act, y: Act, name="pool"):
        self.plan.add(self.lib.ammc_maxpool2x2_s16 if self.s16 else self.lib.ammc_maxpool2x2_f32,
                      x.pix0(), *x.strides, y.pix0(), *y.strides,
                      y.B, y.H, y.W, y.c, name=name, nbytes=4.0 * 5 * y.B * y.H * y.W * y.c, kernel="maxpool2x2")

    def convt(self, x: Act, w: torch.Tensor, bias: torch.Tensor, y: Act, name="convt"):
        co = w.shape[0] // 4
        # bias per GEMM column: the same [co] vector for each of the 4 (dy,dx) groups
        b4 = bias.repeat(4).contiguous()
        self.plan.keep.append(b4)
        self.conv(x, w, y, ntaps=1, cin=w.shape[1], n=4 * co, shift=b4, up=2, cgroup=co, name=name)


class StreamGraph:
    """buffers + plan fragments of one U-Net stream at one input shape"""

    def __init__(self, bld: _Builder, sp: _StreamPack, B: int, H: int, W: int, index: int = 0):
        self.index = index                          # which input tensor of the forward this stream reads
        if H < 8 or W < 8:
            raise ValueError(f"frame size {H}x{W}: three 2x2 poolings need at least 8x8")
        self.sp, self.B, self.H, self.W = sp, B, H, W
        self.bld = bld
        chans = (64, 128, 256, 512)
        # level sizes as nn.MaxPool2d(2) leaves them (floor); where a level is odd, the transposed conv of the decoder
        # gives 2 * floor(h / 2) = h - 1 rows and `up.forward` pads one zero row / column at the END
        # (reference unet.py:53-56: F.pad(x1, (dX // 2, dX - dX // 2, dY // 2, dY - dY // 2)) with dX, dY in {0, 1}):
        # the transposed conv writes at offset 0 of a zero-initialised slice that nothing else writes
        self.hs = [H, H // 2, H // 2 // 2, H // 2 // 2 // 2]
        self.ws = [W, W // 2, W // 2 // 2, W // 2 // 2 // 2]
        self.x_in = bld.act(B, H, W, sp.inc.cin_p)
        # concat buffers of the three decoder levels; first half = the encoder's skip tensor
        self.cat = [bld.act(B, self.hs[i], self.ws[i], 2 * chans[i]) for i in range(3)]
        self.skip = [self.cat[i].slice(0, chans[i]) for i in range(3)]
        self.x4 = bld.act(B, self.hs[3], self.ws[3], 512)
        self.bottom = self.x4          # what the decoder consumes (replaced by vq / bridge outputs)

    def encode(self):
        bld, sp = self.bld, self.sp
        B, H, W = self.B, self.H, self.W
        chans = (64, 128, 256, 512)
        # every encoder level stores its skip tensor and, for the `down` block that follows (unet.py:36), the 2x2
        # max-pool of it: from the conv's own epilogue where the halo-patch kernel runs the layer, else by a pool launch
        mid = bld.act(B, H, W, 64)
        pooled = bld.act(B, self.hs[1], self.ws[1], chans[0])
        # the first layer reads the NCHW input itself where csrc/conv_first_s16.hip applies (launched per forward by
        # EvalEngine._launch_all: the input pointer changes); else layout kernel + implicit GEMM
        self.first_mid = mid if (bld.s16 and sp.first is not None and W % 32 == 0 and H % 8 == 0 and
                                 os.environ.get("AMMC_FIRST_FUSED", "1") != "0") else None
        if self.first_mid is not None:
            p, si = sp.inc, self.index

            def first_args(ctx, m=mid, p=p, si=si, B=B, H=H, W=W, cin=sp.cin, img=sp.first, flag=bld.overflow):
                x = ctx["x"][si]              # (batch stride: clips may be overlapping windows of a resident sub-video)
                return (_ptr(x), x.stride(0), B, cin, H, W, _ptr(img), _ptr(p.s0), _ptr(p.b0), ACT_RELU, m.pix0(), *m.strides,
                        flag.data_ptr())
            bld.plan.keep.extend([sp.first, p.s0, p.b0])
            bld.plan.add(bld.lib.ammc_conv_first_s16_bs, name="inc.conv0", kernel="conv_first_s16", dyn=first_args,
                         flops=2.0 * B * H * W * 9 * sp.cin * 64, nbytes=4.0 * B * H * W * (sp.cin + 64))
            # (Measured and dropped: running this layer and the next image chunk by image chunk, so that the 268-MB
            # intermediate is read back from the Infinity Cache instead of HBM - 4 images per launch 6.73 -> 6.85 ms per step,
            # 2 images 6.99: the smaller launches lose more than the cache gives; these layers are not HBM-bandwidth-bound.)
            fused = bld.double_conv(self.x_in, sp.inc, mid, self.skip[0], name="inc", pool=pooled, skip_first=True)
        else:
            fused = bld.double_conv(self.x_in, sp.inc, mid, self.skip[0], name="inc", pool=pooled)
        for i in range(3):
            h, w = self.hs[i + 1], self.ws[i + 1]
            if not fused:
                bld.maxpool(self.skip[i], pooled, name=f"down{i + 1}.pool")
            mid = bld.act(B, h, w, chans[i + 1])
            out = self.skip[i + 1] if i < 2 else self.x4
            nxt = bld.act(B, self.hs[i + 2], self.ws[i + 2], chans[i + 1]) if i < 2 else None
            fused = bld.double_conv(pooled, sp.down[i], mid, out, name=f"down{i + 1}", pool=nxt)
            pooled = nxt

    def memory(self):
        """enc 1x1 -> fused distance/top-k/gather -> dec 1x1 + residual  (unet.py:318-331, 379-387)"""
        bld, v = self.bld, self.sp.vq
        B, h, w = self.B, self.hs[3], self.ws[3]
        n, d, m, k = B * h * w, v["d"], v["m"], v["k"]
        lib = bld.lib
        self.idx = torch.zeros((n, k), device=bld.device, dtype=torch.int32)
        self.q_one = bld.buf(B, h, w, d)
        nblk = lib.ammc_memory_topk_blocks(n)
        self.diff_part = bld.buf(nblk)
        self.diff = bld.buf(1)
        bld.plan.keep.extend([self.idx, v["embed"], v["e_md"], v["enorm"]])
        if (bld.s16 and v.get("e_s16") is not None and FUSED_MEMORY and (d, k) == (64, 2) and m <= 2048 and
                self.x4.c == 512 and self.x4.c_off == 0 and os.environ.get("AMMC_MEMORY_S16", "1") != "0"):
            # round 6: the whole block as ONE launch (csrc/memory_topk_s16.hip, `memory_block_s16_kernel`): z, the n x m
            # distances, the gathered rows and their S16 re-encoding never leave the CU; bit-identical to the chain below
            self.x4q = bld.act(B, h, w, 512)
            counter = torch.zeros(1, device=bld.device, dtype=torch.int32)      # the last workgroup leaves it at zero
            if v.get("dec_wf") is None:                  # dec's filters in fragment order (once per parameter version)
                v["dec_wf"] = torch.empty_like(v["dec_w"])
                _lib.check(lib.ammc_pack_frag_rows_s16(_ptr(v["dec_w"]), 512, k * d, _ptr(v["dec_wf"]),
                                                       torch.cuda.current_stream(bld.device).cuda_stream), "pack_frag_rows")
            bld.plan.keep.extend([v["e_s16"], v["enc_w"], v["enc_b"], v["dec_wf"], v["dec_b"], counter])
            bld.plan.add(lib.ammc_memory_block_s16, self.x4.pix0(), *self.x4.strides, self.x4q.pix0(), *self.x4q.strides,
                         B, h, w, 512, _ptr(v["enc_w"]), _ptr(v["enc_b"]), v["e_s16"].data_ptr(), _ptr(v["e_md"]),
                         _ptr(v["enorm"]), d, m, k, _ptr(v["dec_wf"]), _ptr(v["dec_b"]), self.idx.data_ptr(), None,
                         _ptr(self.q_one), _ptr(self.diff_part), _ptr(self.diff), counter.data_ptr(),
                         bld.overflow.data_ptr(), name="vq.block",
                         flops=2.0 * n * (512 * d + d * m + k * d * 512), nbytes=4.0 * n * (512 + 512 + 512 + k + d),
                         kernel="memory_block_s16")
            self.bottom = self.x4q
            return
        self.z = bld.act(B, h, w, d, halo=0)
        bld.conv(self.x4, v["enc_w"], self.z, ntaps=1, cin=512, n=d, shift=v["enc_b"], name="vq.enc", y_f32=True)
        self.qk = bld.act(B, h, w, k * d, halo=0)
        if v.get("e_s16") is not None and os.environ.get("AMMC_MEMORY_S16", "1") != "0":
            # S16 plans: the distance GEMM in fp32-equivalent split-fp16 arithmetic (csrc/memory_topk_s16.hip)
            bld.plan.keep.append(v["e_s16"])
            bld.plan.add(lib.ammc_memory_topk_fwd_s16, _ptr(self.z.buf), v["e_s16"].data_ptr(), _ptr(v["e_md"]),
                         _ptr(v["enorm"]), n, d, m, k, self.idx.data_ptr(), _ptr(self.qk.buf), _ptr(self.q_one),
                         _ptr(self.diff_part), name="vq.memory_topk", flops=2.0 * n * d * m,
                         nbytes=4.0 * (n * d + d * m + n * k * d + n * k + n * d), kernel="memory_topk_s16")
        else:
            bld.plan.add(lib.ammc_memory_topk_fwd_f32, _ptr(self.z.buf), _ptr(v["embed"]), _ptr(v["e_md"]),
                         _ptr(v["enorm"]), n, d, m, k, self.idx.data_ptr(), _ptr(self.qk.buf), _ptr(self.q_one),
                         _ptr(self.diff_part), name="vq.memory_topk", flops=2.0 * n * d * m,
                         nbytes=4.0 * (n * d + d * m + n * k * d + n * k + n * d), kernel="memory_topk")
        bld.plan.add(lib.ammc_sum_partials_f32, _ptr(self.diff_part), nblk, 1.0 / float(n * d), _ptr(self.diff),
                     name="vq.diff")
        self.x4q = bld.act(B, h, w, 512)
        if bld.s16:              # the gathered fp32 rows become the S16 operand of `dec`
            qk_s = bld.act(B, h, w, k * d, halo=0)
            # (guarded: a gathered slot beyond the half range - an un-hit slot of a from-scratch codebook picked as second
            # neighbour - raises the plan's range flag HERE, not only if the `dec` epilogue downstream happens to see inf)
            bld.plan.add(lib.ammc_split_rows_guarded_f32, _ptr(self.qk.buf), n * k * d, _ptr(qk_s.buf), bld.overflow.data_ptr(),
                         name="vq.split", nbytes=8.0 * n * k * d, kernel="split_rows")
            self.qk_op = qk_s
        else:
            self.qk_op = self.qk
        bld.conv(self.qk_op, v["dec_w"], self.x4q, ntaps=1, cin=k * d, n=512, shift=v["dec_b"], res=self.x4,
                 name="vq.dec")
        self.bottom = self.x4q

    def decode(self):
        bld, sp = self.bld, self.sp
        B, H, W = self.B, self.H, self.W
        chans = (64, 128, 256, 512)
        y = self.bottom
        for j, lvl in enumerate((2, 1, 0)):            # up1 -> level 2 (H/4), up2 -> level 1, up3 -> level 0
            wt, bias, dc = sp.up[j]
            c = chans[lvl]
            h, w = self.hs[lvl], self.ws[lvl]
            mid = bld.act(B, h, w, c)
            out = bld.act(B, h, w, c)
            if sp.up_fused[j] is not None and bld.up_conv_eligible(self.skip[lvl], c) and (h, w) == (2 * y.H, 2 * y.W):
                # transposed conv + concat + first conv as one launch; the up half of the concat buffer stays unused
                bld.up_conv(y, self.skip[lvl], dc, sp.up_fused[j], mid, name=f"up{j + 1}.up+conv0")
                bld.conv(mid, dc.w1, out, ntaps=9, cin=c, n=c, scale=dc.s1, shift=dc.b1, act=ACT_RELU, name=f"up{j + 1}.conv1")
            else:
                bld.convt(y, wt, bias, self.cat[lvl].slice(c, c), name=f"up{j + 1}.up")
                bld.double_conv(self.cat[lvl], dc, mid, out, name=f"up{j + 1}")
            y = out
        self.u3 = y
        self.outc = bld.outc_desc(y, sp.outc_w, sp.outc_b, sp.cout)
        self.outc_kernel = bld.outc_kernel


class EvalEngine:
    """eval-mode forward of `UNet`, `UNetMem_v7` or `twostream` on the HIP kernels"""

    def __init__(self, module, kind: str, precision: str = "fp32"):
        if precision not in ("fp32", "s16"):
            raise ValueError("precision must be 'fp32' (exact fp32 MFMA) or 's16' (split-fp16 MFMA, fp32-equivalent)")
        self.module = module
        self.kind = kind                      # "unet" | "unetmem" | "twostream"
        self.precision = precision
        self.s16 = precision == "s16"
        self.lib = _lib.load()
        self._packs = None
        self._pack_version = None
        self._plans: Dict[Tuple, dict] = {}
        self._arenas: Dict[Tuple, Tuple[int, List[torch.Tensor]]] = {}
        self._timed = False          # bench.py: bracket every launch with HIP events
        self.memory_fp32_routed = 0  # codebooks of the current packs that do not fit the half range (looked up in fp32)
        self.weights_out_of_range = False
        self.use_graph = USE_GRAPH   # replay one captured hipGraph per forward (AMMC_GRAPH=0: ~100 eager launches)
        self.timings = []

    # ---- parameters ---------------------------------------------------------------
    def _version(self):
        v = 0
        for t in self.module.parameters():
            v += t._version
        for t in self.module.buffers():
            v += t._version
        first = next(self.module.parameters())
        # _param_epoch: bumped by the training engine, which updates buffers through raw pointers
        return (v, first.device, first.data_ptr(), getattr(self.module, "_param_epoch", 0))

    def _ensure_packs(self, device):
        ver = self._version()
        if self._packs is not None and ver == self._pack_version:
            return
        pk = _Packer(device, self.s16)
        m = self.module
        if self.kind == "twostream":
            self._packs = dict(rgb=_StreamPack(pk, m.rgb), op=_StreamPack(pk, m.op),
                               o2f=_DoubleConvPack(pk, m.bridge.O2F), f2o=_DoubleConvPack(pk, m.bridge.F20))
        else:
            self._packs = dict(net=_StreamPack(pk, m))
        self._pack_version = ver
        self._plans.clear()               # plans hold pointers to the old packs
        self._arenas.clear()
        # Range verdicts of the S16 packs: ONE small read per parameter version (packing is rare in evaluation).
        #  * a codebook with an entry beyond the half range (the reference's from-scratch EMA state: every slot no row has hit
        #    yet, ~1e5 x N(0, 1) for the first ~150 training steps) is looked up by the fp32 kernel - the memory block only
        #    (0.07 of 6.3 ms per step): nothing else of the S16 plan changes and no batch is re-run;
        #  * a FILTER beyond the half range cannot be represented at all: every forward reports overflow and the guard
        #    evaluates on the exact-fp32 plans (`overflowed` / `take_overflow`).
        self.memory_fp32_routed = 0
        self.weights_out_of_range = False
        if pk.flags is not None:
            verdicts = pk.flags.tolist()
            self.weights_out_of_range = bool(verdicts[0])
            for sp in self._packs.values():
                vq = getattr(sp, "vq", None)
                if vq and vq.get("e_s16_flag") is not None and verdicts[vq["e_s16_flag"]]:
                    vq["e_s16"] = None
                    self.memory_fp32_routed += 1

    # ---- plans ----------------------------------------------------------------------
    def _build(self, B, H, W, device) -> dict:
        plan = Plan()
        akey = (H, W, device)
        arena = self._arenas.get(akey)
        bld = _Builder(plan, device, self.s16, arena[1] if arena is not None and arena[0] >= B else None)
        st = {}
        if self.kind == "twostream":
            r = StreamGraph(bld, self._packs["rgb"], B, H, W, index=0)
            o = StreamGraph(bld, self._packs["op"], B, H, W, index=1)
            # (plan.lane: which of the two network streams a call belongs to - AMMC_EVAL_LANES runs them on two HIP streams)
            plan.lane = 0
            r.encode()
            r.memory()
            plan.lane = 1
            o.encode()
            o.memory()
            h, w = r.hs[3], r.ws[3]
            # AMFT bridge: x = zx + O2F(zy); y = zy + F20(zx)   (unet.py:962-965): each half reads BOTH bottlenecks
            mid = bld.act(B, h, w, 512)
            xb = bld.act(B, h, w, 512)
            plan.lane = 2                                  # lane 0 behind a join of the two lanes
            bld.double_conv(o.x4q, self._packs["o2f"], mid, xb, res=r.x4q, name="bridge.O2F")
            mid2 = bld.act(B, h, w, 512)
            yb = bld.act(B, h, w, 512)
            plan.lane = 3                                  # lane 1 behind the same join
            bld.double_conv(r.x4q, self._packs["f2o"], mid2, yb, res=o.x4q, name="bridge.F20")
            r.bottom, o.bottom = xb, yb
            plan.lane = 0
            r.decode()
            plan.lane = 1
            o.decode()
            plan.lane = None
            st = dict(streams=[r, o], bridge=(xb, yb))
        else:
            s = StreamGraph(bld, self._packs["net"], B, H, W)
            s.encode()
            if self.kind == "unetmem":
                s.memory()
            s.decode()
            st = dict(streams=[s])
        st["plan"] = plan
        st["overflow"] = bld.overflow
        st["splitk"] = bld.splitk
        if arena is None or arena[0] < B:
            self._arenas[akey] = (B, bld.made)
        return st

    def _get(self, B, H, W, device) -> dict:
        self._ensure_packs(device)
        key = (B, H, W, device)
        st = self._plans.get(key)
        if st is None:
            st = self._build(B, H, W, device)
            self._plans[key] = st
        return st

    def bottleneck_views(self):
        """NCHW views of the rgb bottleneck before / after the memory block (the reference's
        `quant_befor` / `quant_after`, unet.py:986,988).  Views of the workspace: valid until the
        next forward of the same shape."""
        s = self._last["streams"][0]
        return self.act_nchw(s.x4), self.act_nchw(s.x4q)

    def overflowed(self, reset: bool = True) -> bool:
        """did an S16 activation of the last forward leave the half range?  (one device sync)"""
        flag = self._last.get("overflow")
        if flag is None:
            return False
        if self.weights_out_of_range:          # a filter that has no S16 image: every batch goes to the fp32 plans
            return True
        ev = self._last.get("flag_event")
        if ev is not None:                     # copied to pinned memory ahead of the output layers (`_launch_all`)
            ev.synchronize()
            hit = bool(int(self._last["flag_host"][0]))
        else:
            hit = bool(flag.item())
        if hit and reset:
            flag.zero_()
        return hit

    def take_overflow(self) -> Optional[torch.Tensor]:
        """the overflow flag of the last forward as a [1] float tensor on the device, and the sticky flag cleared -
        both queued on the stream, nothing waits (for harness loops that read many batches' flags at once)"""
        flag = self._last.get("overflow")
        if flag is None:
            return None
        out = flag.to(torch.float32) if not self.weights_out_of_range else torch.ones(1, device=flag.device)
        flag.zero_()
        return out

    def act_nchw(self, a: Act) -> torch.Tensor:
        """an activation of the workspace as an NCHW fp32 tensor (a view for fp32 plans, a decoded copy
        for S16 plans); halo-free activations (z, qk) are fp32 in both"""
        if not self.s16 or a.halo == 0:
            return a.interior().permute(0, 3, 1, 2)
        y = torch.empty((a.B, a.c, a.H, a.W), device=a.buf.device, dtype=torch.float32)
        _lib.check(self.lib.ammc_s16_to_nchw_f32(a.pix0(), *a.strides, a.B, a.c, a.H, a.W, _ptr(y),
                                                 torch.cuda.current_stream(y.device).cuda_stream), "s16_to_nchw")
        return y

    # ---- forward ----------------------------------------------------------------------
    def _launch_all(self, st, B, H, W, xs, ys, tgts, accs, stream, launch, early_flag: bool = False):
        """every launch of one forward, in order: input layout, the plan, the two `outc` layers.
        `early_flag` (eager S16 forwards): the range flag is copied to pinned host memory BEFORE the two output
        layers are launched - they write fp32 frames and cannot raise it - so that the guard's wait (`overflowed`) ends
        while the device still has ~0.25 ms of work queued: the host's preparation of the next forward overlaps it
        instead of leaving the device idle (reading the flag after the last launch cost 0.29 ms per forward, 4 %)."""
        lib = self.lib
        streams: List[StreamGraph] = st["streams"]
        ctx = dict(x=xs)
        for s, x in zip(streams, xs):
            if getattr(s, "first_mid", None) is not None:
                continue                                    # the first layer reads the NCHW tensor itself (plan entries)
            launch(lib.ammc_nchw_to_s16_f32 if self.s16 else lib.ammc_nchw_to_nhwc_f32,
                   (_ptr(x), B, s.sp.cin, H, W, s.x_in.pix0(), *s.x_in.strides, s.sp.inc.cin_p),
                   dict(name="nchw_to_nhwc", kernel="nchw_to_nhwc", flops=0.0,
                        bytes=4.0 * B * H * W * (s.sp.cin + s.sp.inc.cin_p)))
        plan = st["plan"]
        lanes = getattr(self, "_lanes", None) if (EVAL_LANES and len(streams) == 2 and not self._timed and
                                                    not torch.cuda.is_current_stream_capturing()) else None
        if EVAL_LANES and lanes is None and len(streams) == 2 and not self._timed and not torch.cuda.is_current_stream_capturing():
            dev = xs[0].device
            lanes = self._lanes = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        if lanes is not None:
            # the two network streams on two HIP streams: forked from the caller's stream, crossed once in front of the
            # bridge (each half of it reads both bottlenecks), joined behind the decoders
            cur = torch.cuda.current_stream(xs[0].device)
            for l in lanes:
                l.wait_stream(cur)
            crossed = False
            try:
                for (fn, args, name), meta in zip(plan.calls, plan.meta):
                    ln = meta.get("lane")
                    if ln is None:
                        raise RuntimeError("a two-stream plan holds a call without a lane")
                    if ln >= 2 and not crossed:
                        lanes[0].wait_stream(lanes[1])
                        lanes[1].wait_stream(lanes[0])
                        crossed = True
                    rc = fn(*(args(ctx) if callable(args) else args), lanes[ln & 1].cuda_stream)
                    if rc != 0:
                        _lib.check(rc, meta["name"])
            finally:
                # joined also when a launch was refused mid-plan: the lanes' queued work uses workspace buffers that the
                # caller's next forward (or the fp32 fallback) touches on its own stream
                for l in lanes:
                    cur.wait_stream(l)
        else:
            for (fn, args, _), meta in zip(plan.calls, plan.meta):
                launch(fn, args(ctx) if callable(args) else args, meta)
        st["flag_event"] = None
        if early_flag and st.get("overflow") is not None:
            if st.get("flag_host") is None:
                st["flag_host"] = torch.zeros(1, dtype=torch.int32).pin_memory()
                st["flag_ev"] = torch.cuda.Event()
            st["flag_host"].copy_(st["overflow"], non_blocking=True)
            st["flag_ev"].record()
            st["flag_event"] = st["flag_ev"]
        for s, y, tgt, acc in zip(streams, ys, tgts, accs):
            s.outc.y = _ptr(y)
            if tgt is not None:
                s.outc.sq_target, s.outc.sq_acc = _ptr(tgt), _ptr(acc)
            else:
                s.outc.sq_target, s.outc.sq_acc = None, None
            launch(lib.ammc_conv_gemm_s16 if self.s16 else lib.ammc_conv_gemm_f32, (C.byref(s.outc),),
                   dict(name="outc_tanh", kernel=s.outc_kernel,
                        flops=2.0 * B * H * W * 9 * 64 * s.sp.cout, bytes=4.0 * B * H * W * (64 + s.sp.cout)))

    def _graph_for(self, st, B, H, W, device, tflags):
        """hipGraph of one forward over static input / output buffers (captured once per plan and target
        pattern): a replay is ONE host call instead of ~100 ctypes launches, which is what bounds small batches."""
        graphs = st.setdefault("graphs", {})
        g = graphs.get(tflags)
        if g is not None:
            return g
        streams = st["streams"]
        g = dict(x=[torch.zeros(B, s.sp.cin, H, W, device=device) for s in streams],
                 y=[torch.empty(B, s.sp.cout, H, W, device=device) for s in streams])
        g["t"] = [torch.zeros(B, s.sp.cout, H, W, device=device) if f else None for s, f in zip(streams, tflags)]
        g["acc"] = [torch.zeros(B, device=device) if f else None for f in tflags]

        def run_on(stream_handle):
            def launch(fn, args, meta):
                rc = fn(*args, stream_handle)
                if rc != 0:
                    _lib.check(rc, meta["name"])
            for a in g["acc"]:
                if a is not None:
                    a.zero_()
            self._launch_all(st, B, H, W, g["x"], g["y"], g["t"], g["acc"], stream_handle, launch)

        cur = torch.cuda.current_stream(device)
        side = torch.cuda.Stream(device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                    # eager warm-up: one-off function attributes, lazy module load
            run_on(side.cuda_stream)
        cur.wait_stream(side)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run_on(torch.cuda.current_stream(device).cuda_stream)
        g["graph"] = graph
        graphs[tflags] = g
        return g

    def forward(self, *inputs: torch.Tensor, targets=None):
        """`targets`: optional per-stream NCHW tensors shaped like the predicted frames (None entries allowed);
        the `outc` epilogue then also accumulates the per-sample squared error of `psnr_error`
        (utils/utils.py:141-148) and `self.last_psnr` holds one [B] tensor (or None) per stream."""
        x0 = inputs[0]
        if not x0.is_cuda:
            raise _lib.AmmcHipError("the HIP path needs CUDA/HIP tensors (module and inputs on the GPU); "
                                    "there is no CPU fallback")
        B, _, H, W = x0.shape
        st = self._get(B, H, W, x0.device)
        stream = torch.cuda.current_stream(x0.device).cuda_stream
        streams: List[StreamGraph] = st["streams"]
        timed = self._timed
        recs = []
        tg = []
        for si, (s, x) in enumerate(zip(streams, inputs)):
            if x.shape[1] != s.sp.cin or x.shape[0] != B or x.shape[2] != H or x.shape[3] != W:
                raise ValueError(f"input shape {tuple(x.shape)} does not match the model ({s.sp.cin} channels)")
            tgt = targets[si] if targets is not None and si < len(targets) else None
            if tgt is not None and tuple(tgt.shape) != (B, s.sp.cout, H, W):
                raise ValueError(f"target shape {tuple(tgt.shape)} != prediction shape {(B, s.sp.cout, H, W)}")
            tg.append(tgt)

        use_graph = self.use_graph and not timed and not torch.cuda.is_current_stream_capturing()
        if use_graph:
            g = self._graph_for(st, B, H, W, x0.device, tuple(t is not None for t in tg))
            for dst, x in zip(g["x"], inputs):
                dst.copy_(x.detach())
            for dst, t in zip(g["t"], tg):
                if t is not None:
                    dst.copy_(t.detach())
            g["graph"].replay()
            st["flag_event"] = None              # replays copy nothing to the host: `overflowed` reads the device flag itself
            outs = [y.clone() for y in g["y"]]
            sq = [a.clone() if a is not None else None for a in g["acc"]]
        else:
            def launch(fn, args, meta):
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                rc = fn(*args, stream)
                if timed:
                    e1.record()
                    recs.append((meta, e0, e1))
                if rc != 0:
                    _lib.check(rc, meta["name"])

            xs, keep = [], []
            for s, x in zip(streams, inputs):
                x = x.detach()
                # a batch of clips cut out of a resident sub-video as overlapping windows (harness.clip_windows: planes
                # contiguous, any batch stride) is read in place by the first-layer kernel; everything else is gathered
                windowed = (x.dtype == torch.float32 and getattr(s, "first_mid", None) is not None and x.dim() == 4 and
                            x.stride()[1:] == (H * W, W, 1) and x.stride(0) >= 0)
                if not windowed and (x.dtype != torch.float32 or not x.is_contiguous()):
                    x = x.float().contiguous()
                xs.append(x)
            outs = [torch.empty((B, s.sp.cout, H, W), device=x0.device, dtype=torch.float32) for s in streams]
            tt = [t.detach().float().contiguous() if t is not None else None for t in tg]
            sq = [torch.zeros(B, device=x0.device, dtype=torch.float32) if t is not None else None for t in tg]
            # (no pinned-memory copy / event while the caller captures this forward into its own graph)
            self._launch_all(st, B, H, W, xs, outs, tt, sq, stream, launch,
                             early_flag=self.s16 and not torch.cuda.is_current_stream_capturing())
            if timed:
                torch.cuda.synchronize()
                self.timings = [(m, e0.elapsed_time(e1)) for m, e0, e1 in recs]
        self._last = st
        n_el = [float(s.sp.cout * H * W) for s in streams]
        self.last_psnr = [10.0 * torch.log10(n / a) if a is not None else None for a, n in zip(sq, n_el)]
        if self.kind == "unet":
            return outs[0]
        diffs = tuple(s.diff.clone() for s in streams)
        qs = tuple(s.q_one.clone() for s in streams)
        if self.kind == "unetmem":
            return outs[0], diffs[0], qs[0]
        return outs[0], outs[1], diffs, qs
