"""`PixelDiscriminator` (reference Code/models/pix2pix_networks.py:580-631) on the HIP kernels, forward and backward.

The reference builds it as `PixelDiscriminator(out_channel, const.d_num_filters=[128,256,512,512], use_norm=False)`
(models/__init__.py:123-124, 254-255, 323): three Conv2d(k 4, stride 2, padding 2, bias) + LeakyReLU(0.1) and a final
Conv2d(512 -> 1, k 4, stride 1, padding 2); a 256x256 frame gives a [B,1,34,34] patch map.  The training loop calls it
three times per step (train_helper.py:318-328): on the generated frame for the generator's adversarial term, and on
the target and the detached generated frame for the discriminator's own loss.

Here every conv is one launch of the path's implicit-GEMM kernel (`ammc_conv_gemm_f32`, ntaps 16, x_step = stride)
over NHWC activations with a 2-pixel zero halo; bias and LeakyReLU are applied in its epilogue.  The backward is
hand scheduled: weight gradients through `ammc_conv_wgrad_f32` (ntaps 16), the input gradient of a stride-2 layer
as four 2x2-tap convolutions over the output gradient (one per input-pixel parity, written through doubled output
strides), of the stride-1 layer as one 16-tap convolution with the flipped filter.  `DiscFunction` exposes the pair
as one autograd node, differentiable w.r.t. the frame (the generator's adversarial gradient) and the parameters.
Every forward that may be differentiated takes its own workspace slot, so the three calls of a step coexist.

Round 4 (`module.precision`, default "s16"): the FORWARD convolutions run on the split-fp16 kernel
(`ammc_conv_gemm_s16`, ntaps 16: (hi, lo) half pairs, three fp16 MFMAs per product, fp32 accumulation - fp32-equivalent)
with fp32 outputs; each activation is then re-encoded once into the S16 twin the next layer reads (the fp32 tensor stays
for the weight gradient and the LeakyReLU mask of the backward).  The INPUT-GRADIENT convolutions run on the same kernel:
the gradient of a layer's output is brought into the half range by a power of two found on the device
(`ammc_absmax_bits_f32` + `ammc_split_rows_scaled_f32`, as the generator's gradients are), re-encoded once, and the
scale is undone in the epilogue; fp32 outputs.  The weight gradients read the same two twins (`ammc_conv_wgrad_s16`,
4x4 windows, stride 1 | 2: the im2col form of wgrad_s16.hip).  "fp32" = the exact-fp32 MFMA kernels throughout, as in
round 3.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, AmmcConvDesc, AmmcWgradDesc
from .engine import Act, _cin_pad, _kpad, _ptr

SLOPE = 0.1          # nn.LeakyReLU(0.1, True), pix2pix_networks.py:606
HALO = 2             # = the convs' padding
DEFAULT_PRECISION = __import__("os").environ.get("AMMC_DISC_PRECISION", "s16")


def _chk(rc, what):
    if rc != 0:
        _lib.check(rc, what)


def _out_hw(h: int, stride: int) -> int:
    return (h + 2 * 2 - 4) // stride + 1


class _Layer:
    def __init__(self, cin: int, cout: int, stride: int, s16: bool = False):
        self.cin, self.cout, self.stride = cin, cout, stride
        self.cin_p = max(8, _cin_pad(cin)) if s16 else _cin_pad(cin)       # S16 operands: whole groups of 8 channels
        self.n = cout if cout % 64 == 0 else 32            # GEMM columns (the 1-channel head pads to a 32 tile)
        self.kpad = _kpad(16 * self.cin_p)
        self.rows = self.cin_p if self.cin_p % 64 == 0 else 32        # columns of the input-gradient GEMM
        self.gc = max(self.n, 4)                           # channels of this layer's output-gradient buffer
        taps = 4 if stride == 2 else 16
        self.dkpad = _kpad(taps * self.gc)


class _Slot:
    """what one differentiable forward keeps: NHWC activations and the filters as packed at forward time"""

    def __init__(self, eng: "DiscEngine", B: int, H: int, W: int):
        dev = eng.device
        self.B, self.H, self.W = B, H, W
        self.acts: List[Act] = []
        h, w = H, W
        for i, L in enumerate(eng.layers):
            self.acts.append(Act(torch.zeros(B, h + 2 * HALO, w + 2 * HALO, L.cin_p, device=dev), B, h, w, L.cin_p, 0,
                                 HALO))
            h, w = _out_hw(h, L.stride), _out_hw(w, L.stride)
        self.out_hw = (h, w)
        self.wp = [torch.zeros(L.n, L.kpad, device=dev) for L in eng.layers]
        self.bp = [torch.zeros(L.n, device=dev) for L in eng.layers]
        self.wd = [torch.zeros((4 if L.stride == 2 else 1) * L.rows * L.dkpad, device=dev) for L in eng.layers]
        self.has_dgrad = False
        # split-fp16 forward: the S16 twins of the layer inputs and of the packed filters
        self.acts16 = [Act(torch.zeros_like(a.buf), a.B, a.H, a.W, a.c, 0, HALO) for a in self.acts] if eng.s16 else None
        self.w16 = [torch.zeros_like(w) for w in self.wp] if eng.s16 else None
        self.wd16 = [torch.zeros_like(w) for w in self.wd] if eng.s16 else None


class _Lease:
    """returns the slot to the engine's pool when the autograd node dies (after backward, or if it never ran)"""

    def __init__(self, pool: list, slot: _Slot):
        self.pool, self.slot = pool, slot

    def __del__(self):
        self.pool.append(self.slot)


class DiscEngine:
    def __init__(self, module: "PixelDiscriminator", precision: str = "s16"):
        if precision not in ("s16", "fp32"):
            raise ValueError("PixelDiscriminator.precision must be 's16' or 'fp32'")
        self.module = module
        self.s16 = precision == "s16"
        self.lib = _lib.load()
        self.device = None
        self.layers: List[_Layer] = []
        chans = [module.input_nc] + list(module.num_filters[:-1])
        for i in range(len(chans) - 1):
            self.layers.append(_Layer(chans[i], chans[i + 1], 2, self.s16))
        self.layers.append(_Layer(module.num_filters[-1], 1, 1, self.s16))
        self._pools: Dict[Tuple[int, int, int], list] = {}
        self._grads: Dict[Tuple[int, int, int], List[Act]] = {}
        self._grads16: Dict[Tuple[int, int, int], List[Act]] = {}
        self._scratch: Optional[torch.Tensor] = None
        self._zeros: Optional[torch.Tensor] = None
        self.slots_created = 0

    # ---- plumbing --------------------------------------------------------------------------------------------
    @property
    def s(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _setup(self, device):
        if self.device != device:
            self.device = device
            self._pools.clear()
            self._grads.clear()
            self._grads16.clear()
            self._zeros = torch.zeros(1024, device=device)
            self._scratch = None

    def _take(self, B, H, W) -> Tuple[_Slot, list]:
        pool = self._pools.setdefault((B, H, W), [])
        if pool:
            return pool.pop(), pool
        self.slots_created += 1
        return _Slot(self, B, H, W), pool

    def _grad_acts(self, slot: _Slot) -> List[Act]:
        """gradient w.r.t. each layer's output, [B, OH, OW, gc] with a 1-pixel zero halo (shared by all slots)"""
        key = (slot.B, slot.H, slot.W)
        if key not in self._grads:
            gs = []
            for i, L in enumerate(self.layers):
                if i + 1 < len(self.layers):
                    h, w = slot.acts[i + 1].H, slot.acts[i + 1].W
                else:
                    h, w = slot.out_hw
                gs.append(Act(torch.zeros(slot.B, h + 2, w + 2, L.gc, device=self.device), slot.B, h, w, L.gc, 0, 1))
            self._grads[key] = gs
            if self.s16:                           # S16 twins of the gradients + the slots / scale of their rescaling
                self._grads16[key] = [Act(torch.zeros_like(g.buf), g.B, g.H, g.W, g.c, 0, 1) for g in gs]
        return self._grads[key]

    def _conv(self, x_ptr: int, x_strides, w: torch.Tensor, y_ptr: int, y_strides, *, batch, height, width, cin, ntaps,
              n, shift=None, act=ACT_NONE, x_step=1, n_store=0, y_cs=0, what="conv", s16=False, scale=None):
        d = AmmcConvDesc()
        d.x, d.w, d.y = x_ptr, _ptr(w), y_ptr
        d.scale = _ptr(scale) if scale is not None else None
        d.shift, d.res = (_ptr(shift) if shift is not None else None), None
        d.batch, d.height, d.width = batch, height, width
        d.cin, d.ntaps, d.n, d.up, d.cgroup, d.act, d.x_step = cin, ntaps, n, 1, n, act, x_step
        d.n_store, d.y_cs = n_store, y_cs
        d.x_bs, d.x_rs, d.x_ps = x_strides
        d.y_bs, d.y_rs, d.y_ps = y_strides
        if s16:                                     # S16 operands, fp32 output
            d.y_f32 = 1
            _chk(self.lib.ammc_conv_gemm_s16(C.byref(d), self.s), what)
        else:
            _chk(self.lib.ammc_conv_gemm_f32(C.byref(d), self.s), what)

    # ---- forward ---------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, params: Sequence[torch.Tensor], keep: bool, need_dx: bool):
        if not x.is_cuda:
            raise _lib.AmmcHipError("PixelDiscriminator runs on the HIP kernels only: input must be a GPU tensor")
        x = x.contiguous().float()
        B, Cx, H, W = x.shape
        if Cx != self.layers[0].cin:
            raise ValueError(f"expected {self.layers[0].cin} input channels, got {Cx}")
        self._setup(x.device)
        lib, s = self.lib, self.s
        slot, pool = self._take(B, H, W)
        a0 = slot.acts[0]
        _chk(lib.ammc_nchw_to_nhwc_f32(_ptr(x), B, Cx, H, W, a0.pix0(), *a0.strides, a0.c, s), "nchw_to_nhwc")
        for i, L in enumerate(self.layers):
            w, b = params[2 * i], params[2 * i + 1]
            _chk(lib.ammc_pack_conv_weight_f32(_ptr(w), L.cout, L.cin, 4, L.cin_p, _ptr(slot.wp[i]), s), "pack_w")
            slot.bp[i][:L.cout].copy_(b)
            if keep and (i > 0 or need_dx):
                _chk(lib.ammc_pack_conv4_dgrad_weight_f32(_ptr(w), L.cout, L.cin, L.gc, L.rows, L.stride, 2,
                                                          _ptr(slot.wd[i]), s), "pack_dgrad")
                if self.s16:
                    _chk(lib.ammc_split_rows_f32(_ptr(slot.wd[i]), slot.wd[i].numel(), _ptr(slot.wd16[i]), s), "split_rows(wd)")
        slot.has_dgrad = keep
        oh, ow = slot.out_hw
        out = torch.empty(B, 1, oh, ow, device=x.device, dtype=torch.float32)
        s16 = self.s16

        def twin(t32: torch.Tensor, t16: torch.Tensor):
            _chk(lib.ammc_split_rows_f32(_ptr(t32), t32.numel(), _ptr(t16), s), "split_rows")

        if s16:
            twin(a0.buf, slot.acts16[0].buf)
            for i in range(len(self.layers)):
                twin(slot.wp[i], slot.w16[i])
        for i, L in enumerate(self.layers):
            a = slot.acts16[i] if s16 else slot.acts[i]
            wts = slot.w16 if s16 else slot.wp
            x_origin = _ptr(a.buf)                                     # window corner of output pixel (0,0)
            if i + 1 < len(self.layers):
                y = slot.acts[i + 1]
                self._conv(x_origin, a.strides, wts[i], y.pix0(), y.strides, batch=B, height=y.H, width=y.W,
                           cin=L.cin_p, ntaps=16, n=L.n, shift=slot.bp[i], act=ACT_LRELU, x_step=L.stride,
                           what=f"disc.conv{i}", s16=s16)
                if s16:
                    twin(y.buf, slot.acts16[i + 1].buf)                # what the next layer reads
            else:
                self._conv(x_origin, a.strides, wts[i], _ptr(out), (oh * ow, ow, 1), batch=B, height=oh, width=ow,
                           cin=L.cin_p, ntaps=16, n=L.n, shift=slot.bp[i], act=ACT_NONE, x_step=1, n_store=1,
                           y_cs=oh * ow, what="disc.head", s16=s16)
        if not keep:
            pool.append(slot)
            return None, out
        return _Lease(pool, slot), out

    # ---- backward --------------------------------------------------------------------------------------------
    def _chan_sum(self, g: Act, c: int) -> torch.Tensor:
        lib = self.lib
        nb = lib.ammc_chan_reduce_blocks(g.B * g.H * g.W)
        if self._scratch is None or self._scratch.numel() < nb * 1024:
            self._scratch = torch.zeros(nb * 1024, device=self.device)
        _chk(lib.ammc_chan_sum_f32(g.pix0(), *g.strides, g.B, g.H, g.W, c, _ptr(self._scratch), self.s), "chan_sum")
        out = torch.empty(c, device=self.device, dtype=torch.float32)
        _chk(lib.ammc_reduce_partials_f32(_ptr(self._scratch), nb, c, 1.0, _ptr(out), self.s), "reduce_partials")
        return out

    def backward(self, slot: _Slot, dout: torch.Tensor, need_dx: bool, need_dw: bool):
        lib, s, B = self.lib, self.s, slot.B
        gs = self._grad_acts(slot)
        oh, ow = slot.out_hw
        dout = dout.contiguous().float()
        g_last = gs[-1]
        _chk(lib.ammc_nchw_to_nhwc_f32(_ptr(dout), B, 1, oh, ow, g_last.pix0(), *g_last.strides, g_last.c, s),
             "dout->nhwc")
        grads: List[Optional[torch.Tensor]] = [None] * (2 * len(self.layers))
        dx = None
        for i in reversed(range(len(self.layers))):
            L, g, a = self.layers[i], gs[i], slot.acts[i]
            if i + 1 < len(self.layers):                                # LeakyReLU mask from the layer's own output
                y = slot.acts[i + 1]
                _chk(lib.ammc_lrelu_bwd_f32(y.pix0(), *y.strides, g.pix0(), *g.strides, B, g.H, g.W, g.c, SLOPE, s),
                     "lrelu_bwd")
            g16 = inv = None
            if self.s16 and (need_dw or i > 0 or need_dx):
                # the gradient as an S16 operand (shared by the weight- and the input-gradient kernels): max |g| on the
                # device -> a power of two that puts it at 2^10 -> the re-encoding; 2^-k comes back through the epilogues
                g16 = self._grads16[(slot.B, slot.H, slot.W)][i]
                amax = torch.zeros(256, device=self.device, dtype=torch.int32)
                inv = torch.empty(1024, device=self.device, dtype=torch.float32)
                _chk(lib.ammc_absmax_bits_f32(_ptr(g.buf), g.buf.numel(), amax.data_ptr(), s), "absmax(g)")
                _chk(lib.ammc_split_rows_scaled_f32(_ptr(g.buf), g.buf.numel(), _ptr(g16.buf), amax.data_ptr(), _ptr(inv),
                                                    1024, s), "split_rows_scaled(g)")
            if need_dw:
                grads[2 * i + 1] = self._chan_sum(g, g.c if L.cout > 1 else 4)[:L.cout].clone()
                dwp = torch.zeros(L.n, L.kpad, device=self.device)
                d = AmmcWgradDesc()
                d.dw, d.zeros = _ptr(dwp), _ptr(self._zeros)
                d.batch, d.height, d.width = B, g.H, g.W
                d.n, d.cin, d.ntaps, d.a_step = L.n, L.cin_p, 16, L.stride
                d.g_bs, d.g_rs, d.g_ps = g.strides
                d.a_bs, d.a_rs, d.a_ps = a.strides
                if self.s16:                       # S16 twins of the gradient and of the layer input (written by the forward)
                    d.g, d.a = g16.pix0(), _ptr(slot.acts16[i].buf)
                    _chk(lib.ammc_conv_wgrad_s16(C.byref(d), _ptr(inv), s), f"disc.wgrad{i}(s16)")
                else:
                    d.g, d.a = g.pix0(), _ptr(a.buf)
                    _chk(lib.ammc_conv_wgrad_f32(C.byref(d), s), f"disc.wgrad{i}")
                dw = torch.empty(L.cout, L.cin, 4, 4, device=self.device)
                _chk(lib.ammc_unpack_conv_wgrad_f32(_ptr(dwp), L.cout, L.cin, 4, L.cin_p, _ptr(dw), s), "unpack_wgrad")
                grads[2 * i] = dw
            if i == 0 and not need_dx:
                break
            # ---- input gradient ----
            if i > 0:
                tgt = gs[i - 1]
                y0, ybs, yrs, yps, ycs, nstore = tgt.pix0(), tgt.bs, tgt.rs, tgt.ps, 0, 0
                esz = 4
            else:
                dx = torch.empty(B, L.cin, slot.H, slot.W, device=self.device, dtype=torch.float32)
                y0, ybs, yrs, yps = _ptr(dx), L.cin * slot.H * slot.W, slot.W, 1
                ycs, nstore, esz = slot.H * slot.W, L.cin, 4
            gsrc, wdi = (g16, slot.wd16[i]) if self.s16 else (g, slot.wd[i])
            if L.stride == 1:
                # dA[q] = sum_r g[q + 2 - r] W[r]: a 16-tap window starting at g(q - 1) with the flipped filter
                self._conv(gsrc.pix0() - esz * (g.rs + g.ps), g.strides, wdi, y0, (ybs, yrs, yps), batch=B,
                           height=a.H, width=a.W, cin=g.c, ntaps=16, n=L.rows, n_store=nstore, y_cs=ycs,
                           what=f"disc.dgrad{i}", s16=self.s16, scale=inv)
            else:
                per = L.rows * L.dkpad
                for ph in range(4):
                    py, px = ph >> 1, ph & 1
                    hh, ww = (a.H - py + 1) // 2, (a.W - px + 1) // 2
                    if hh <= 0 or ww <= 0:
                        continue
                    self._conv(gsrc.pix0(), g.strides, wdi[ph * per:(ph + 1) * per],
                               y0 + esz * (py * yrs + px * yps), (ybs, 2 * yrs, 2 * yps), batch=B, height=hh, width=ww,
                               cin=g.c, ntaps=4, n=L.rows, n_store=nstore, y_cs=ycs, what=f"disc.dgrad{i}.{ph}",
                               s16=self.s16, scale=inv)
        return dx, grads


class DiscFunction(torch.autograd.Function):
    """(engine, frame, *parameters) -> patch map; differentiable w.r.t. the frame and the parameters"""

    @staticmethod
    def forward(ctx, engine: DiscEngine, x: torch.Tensor, *params):
        needs = ctx.needs_input_grad
        keep = any(needs)
        lease, out = engine.forward(x, params, keep, bool(needs[1]))
        ctx.engine, ctx.lease = engine, lease
        return out

    @staticmethod
    def backward(ctx, dout):
        needs = ctx.needs_input_grad
        lease, ctx.lease = ctx.lease, None
        if lease is None:
            raise RuntimeError("PixelDiscriminator: backward called twice on the same forward")
        dx, grads = ctx.engine.backward(lease.slot, dout, bool(needs[1]), any(needs[2:]))
        red = getattr(ctx.engine.module, "_grad_reducer", None)
        if red is not None and any(g is not None for g in grads):
            red.push([g for g in grads if g is not None])
            red.finish()
        del lease
        return (None, dx, *grads)


class PixelDiscriminator(nn.Module):
    """Same constructor, state_dict keys (net.0 / net.2 / ... .weight/.bias) and forward as the reference class;
    `use_norm=True` (never used by the reference's model factory) is not built."""

    def __init__(self, input_nc: int, num_filters: Sequence[int], use_norm: bool = False, norm_layer=nn.BatchNorm2d):
        super().__init__()
        if use_norm:
            raise NotImplementedError("PixelDiscriminator(use_norm=True) has no HIP path; the reference's model "
                                      "factory builds it with use_norm=False (models/__init__.py:123-124)")
        self.input_nc, self.num_filters = int(input_nc), [int(c) for c in num_filters]
        net: List[nn.Module] = [nn.Conv2d(input_nc, num_filters[0], kernel_size=4, padding=2, stride=2),
                                nn.LeakyReLU(SLOPE, True)]
        for i in range(1, len(num_filters) - 1):
            net.extend([nn.Conv2d(num_filters[i - 1], num_filters[i], 4, 2, 2, bias=True), nn.LeakyReLU(SLOPE, True)])
        net.append(nn.Conv2d(num_filters[-1], 1, 4, 1, 2))
        self.net = nn.Sequential(*net)                     # parameter holders; never called
        self.precision = DEFAULT_PRECISION
        object.__setattr__(self, "_engine", None)

    def _params(self) -> List[torch.Tensor]:
        out = []
        for m in self.net:
            if isinstance(m, nn.Conv2d):
                out += [m.weight, m.bias]
        return out

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if not input.is_cuda:
            raise _lib.AmmcHipError("PixelDiscriminator runs on the HIP kernels only: input must be a GPU tensor")
        if self._engine is None or self._engine.s16 != (self.precision == "s16"):
            object.__setattr__(self, "_engine", DiscEngine(self, self.precision))
        params = self._params()
        s16 = self._engine.s16
        guard = getattr(self, "s16_guard", True)
        object.__setattr__(self, "last_overflow", None)
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in params)):
            out = DiscFunction.apply(self._engine, input, *params)
            if s16 and guard:
                # training: an activation beyond the half range turns the patch map non-finite (inf in the re-encoding of a
                # layer's output, then inf / NaN through every later layer); the verdict stays on the device for the
                # trainer (`harness.train_step_gan` refuses the step on a non-finite loss AND on this flag)
                object.__setattr__(self, "last_overflow", (~torch.isfinite(out.detach()).all()).to(torch.int32).reshape(1))
            return out
        out = self._engine.forward(input, params, False, False)[1]
        if s16 and guard:
            # inference (no autograd graph): S16 range guard as the generator's - one read of the verdict, recomputation
            # on the exact-fp32 kernels when the patch map is not finite.  "defer": leave the verdict on the device.
            bad = ~torch.isfinite(out).all()
            if guard == "defer":
                object.__setattr__(self, "last_overflow", bad.to(torch.int32).reshape(1))
            elif bool(bad):
                if getattr(self, "_engine_fp32", None) is None:
                    object.__setattr__(self, "_engine_fp32", DiscEngine(self, "fp32"))
                out = self._engine_fp32.forward(input, params, False, False)[1]
                object.__setattr__(self, "s16_fallbacks", getattr(self, "s16_fallbacks", 0) + 1)
        return out
