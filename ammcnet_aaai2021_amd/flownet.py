"""`FlowNet2SD` (reference Code/models/flownet2/models.py:9-59 over FlowNetSD.py:7-100, submodules.py:9-45), the frozen
flow estimator behind the flow-consistency term of the generator loss (train_helper.py:309-316), eval-mode forward on the
HIP kernels.  Same constructor, `state_dict` keys (45,371,666 parameters) and call: `net(frames [B,3,2,H,W] in 0..255)`
-> flow `[B,2,H,W]`.  The published checkpoint is not available here; parity is against the reference class itself on
synthetic parameters (tests/golden/flownet2sd_eval.npz).

Every Conv2d(k 3, stride 1|2, bias) + LeakyReLU(0.1) is one launch of `ammc_conv_gemm_f32` (ntaps 9, x_step = stride);
every ConvTranspose2d(k 4, s 2, p 1) is four 2x2-tap parity convolutions through doubled output strides (the forward of a
transposed conv IS the input gradient of a conv: the filters come from `ammc_pack_conv4_dgrad_weight_f32`, pad 1).
Concatenations (`torch.cat`, models.py:35-52) are channel slices of one buffer; a layer that READS a concatenation runs
once per part (channel counts 1026 / 770 / 386 / 194 are not what the kernel tiles; their parts 512+512+2 ... are) and
accumulates through the epilogue's residual input.

Two arithmetic forms (`net.precision`): "s16" (default, round 4) = the split-fp16 kernels the generator runs on
(`ammc_conv_gemm_s16`: (hi, lo) half pairs, three fp16 MFMAs per product, fp32 accumulation - fp32-equivalent; flows of
tens of pixels are far inside the half range), activations and packed filters kept as S16; the 2-channel flow heads and
their 2 -> 2 up-convolutions stay fp32 (their outputs enter the concatenations as 8-channel S16 parts).  "fp32" = the
exact-fp32 MFMA kernels throughout (round 3: 13.5 ms per batch-32 forward against ~3 for S16).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, AmmcConvDesc
from .engine import Act, _cin_pad, _kpad, _ptr

SLOPE = 0.1
DEFAULT_PRECISION = __import__("os").environ.get("AMMC_FLOWNET_PRECISION", "s16")


def _chk(rc, what):
    if rc != 0:
        _lib.check(rc, what)


def _ncols(c: int) -> int:
    return c if c % 64 == 0 else 32


class _Engine:
    def __init__(self, module: "FlowNet2SD", precision: str = "s16"):
        if precision not in ("s16", "fp32"):
            raise ValueError("FlowNet2SD.precision must be 's16' or 'fp32'")
        self.m = module
        self.s16 = precision == "s16"
        self.lib = _lib.load()
        self.ws: Dict[Tuple, dict] = {}
        self.packs: Optional[dict] = None
        self.pack_version = None

    # ---- parameters ------------------------------------------------------------------------------------------
    def _version(self):
        ps = list(self.m.parameters())
        return (sum(p._version for p in ps), ps[0].device, ps[0].data_ptr())

    def _cp(self, c: int, fp32_layer: bool = False) -> int:
        """padded channel count of an input part: S16 operands come in whole groups of 8"""
        return max(8, _cin_pad(c)) if (self.s16 and not fp32_layer) else _cin_pad(c)

    def _w16(self, wp: torch.Tensor, fp32_layer: bool = False) -> torch.Tensor:
        """the packed filter as the layer's kernel reads it: S16 for the split-fp16 layers"""
        if not self.s16 or fp32_layer:
            return wp
        w16 = torch.empty_like(wp)
        _chk(self.lib.ammc_split_rows_f32(_ptr(wp), wp.numel(), _ptr(w16), torch.cuda.current_stream(wp.device).cuda_stream), "split_rows")
        return w16

    def _pack_conv(self, w: torch.Tensor, b: torch.Tensor, parts: Sequence[int], head: bool = False):
        """3x3 conv over a (possibly concatenated) input: one packed filter per part, bias padded to the column count.
        `head`: a layer that stays on the fp32 kernels in both forms (never set for convs: the flow heads READ S16)"""
        lib, dev = self.lib, w.device
        s = torch.cuda.current_stream(dev).cuda_stream
        co = w.shape[0]
        n = _ncols(co)
        out, c0 = [], 0
        for c in parts:
            cp = self._cp(c, head)
            wp = torch.zeros(n, _kpad(9 * cp), device=dev)
            ws = w.detach()[:, c0:c0 + c].contiguous()
            _chk(lib.ammc_pack_conv_weight_f32(_ptr(ws), co, c, 3, cp, _ptr(wp), s), "pack_conv")
            out.append((self._w16(wp, head), cp))
            c0 += c
        bias = torch.zeros(n, device=dev)
        bias[:co].copy_(b.detach())
        return dict(w=out, bias=bias, n=n, co=co)

    def _pack_deconv(self, w: torch.Tensor, b: torch.Tensor, parts: Sequence[int], head: bool = False):
        """ConvTranspose2d(k 4, s 2, p 1), weight [Cin, Cout, 4, 4]: per input part four parity filters [rows][4*cin_p].
        `head`: the 2 -> 2 up-convolution of a flow estimate (fp32 kernels in both forms)"""
        lib, dev = self.lib, w.device
        s = torch.cuda.current_stream(dev).cuda_stream
        co = w.shape[1]
        rows = _ncols(co)
        out, c0 = [], 0
        wd = w.detach().contiguous()
        for c in parts:
            cp = self._cp(c, head)
            kp = _kpad(4 * cp)
            wp = torch.zeros(4 * rows * kp, device=dev)
            _chk(lib.ammc_pack_conv4_dgrad_weight_f32(_ptr(wd, c0 * co * 16), c, co, cp, rows, 2, 1, _ptr(wp), s),
                 "pack_deconv")
            out.append((self._w16(wp, head), cp, rows * kp))
            c0 += c
        bias = torch.zeros(rows, device=dev)
        bias[:co].copy_(b.detach())
        return dict(w=out, bias=bias, n=rows, co=co, head=head)

    def _ensure_packs(self):
        v = self._version()
        if self.packs is not None and v == self.pack_version:
            return
        m, pk = self.m, {}
        single = ["conv0", "conv1", "conv1_1", "conv2", "conv2_1", "conv3", "conv3_1", "conv4", "conv4_1", "conv5",
                  "conv5_1", "conv6", "conv6_1"]
        for name in single:
            conv = getattr(m, name)[0]
            pk[name] = self._pack_conv(conv.weight, conv.bias, [conv.weight.shape[1]])
        cat = {5: (512, 512, 2), 4: (512, 256, 2), 3: (256, 128, 2), 2: (128, 64, 2)}
        for lvl, parts in cat.items():
            ic = getattr(m, f"inter_conv{lvl}")[0]
            pk[f"inter_conv{lvl}"] = self._pack_conv(ic.weight, ic.bias, parts)
        pk["deconv5"] = self._pack_deconv(m.deconv5[0].weight, m.deconv5[0].bias, (1024,))
        for lvl in (4, 3, 2):
            dc = getattr(m, f"deconv{lvl}")[0]
            pk[f"deconv{lvl}"] = self._pack_deconv(dc.weight, dc.bias, cat[lvl + 1])
        for lvl in (6, 5, 4, 3, 2):
            pf = getattr(m, f"predict_flow{lvl}")
            pk[f"predict_flow{lvl}"] = self._pack_conv(pf.weight, pf.bias, [pf.weight.shape[1]])
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            up = getattr(m, f"upsampled_flow{a}_to_{b}")
            pk[f"up{a}"] = self._pack_deconv(up.weight, up.bias, (2,), head=True)
        self.packs, self.pack_version = pk, v

    # ---- workspace -------------------------------------------------------------------------------------------
    def _workspace(self, B, H, W, dev) -> dict:
        key = (B, H, W, dev)
        ws = self.ws.get(key)
        if ws is not None:
            return ws

        def act(div, c):
            h, w = H // div, W // div
            return Act(torch.zeros(B, h + 2, w + 2, c, device=dev), B, h, w, c, 0, 1)

        ws = dict(x0=act(1, 8), c0=act(1, 64), t1=act(2, 64), c1=act(2, 128), t2=act(4, 128), t3=act(8, 256),
                  t4=act(16, 512), t5=act(32, 512), t6=act(64, 1024), c6=act(64, 1024),
                  # (S16: pixel strides are whole groups of 8 channels - the 2-channel flow part of a concatenation lives in
                  # its own 8-channel buffer, u*, below)
                  cat2=act(4, 192 if self.s16 else 196), cat3=act(8, 384 if self.s16 else 388),
                  cat4=act(16, 768 if self.s16 else 772), cat5=act(32, 1024 if self.s16 else 1028),
                  ic5=act(32, 512), ic4=act(16, 256), ic3=act(8, 128), ic2=act(4, 64),
                  f6=act(64, 4), f5=act(32, 4), f4=act(16, 4), f3=act(8, 4), f2=act(4, 4))
        if self.s16:
            # the up-sampled flow estimates: fp32 from their 2 -> 2 transposed conv (uf*), then the 8-channel S16 part
            # of the concatenation (u*: channels 2..7 zero); the prepared frame pair likewise (x0 fp32 -> x0s)
            ws.update(uf6=act(32, 8), uf5=act(16, 8), uf4=act(8, 8), uf3=act(4, 8),
                      u6=act(32, 8), u5=act(16, 8), u4=act(8, 8), u3=act(4, 8), x0s=act(1, 8))
            # split-K workspace of ammc_conv_gemm_s16: the layers below 1/16 resolution have 4-64 output tiles for 256
            # CUs and K up to 9216 - one workgroup per tile walks 288 chunks alone (conv5_1: 306 us for 32 workgroups);
            # with the workspace the kernel cuts K into up to 16 slices per tile and a streaming kernel finishes
            # (largest user: [16 slices][B * 8 * 8 pixels of conv5_1 at 256x256][512]; 32 M floats cover batch 64)
            ws["splitk"] = torch.empty(max(1 << 22, 16 * B * (H // 32) * (W // 32) * 512), device=dev)
            # sticky range flag of the S16 epilogues (an activation beyond 65504 becomes inf in its hi half): raised on the
            # device, read once per forward by FlowNet2SD.forward, which then recomputes on the exact-fp32 kernels
            ws["flag"] = torch.zeros(1, device=dev, dtype=torch.int32)
        self.ws[key] = ws
        return ws

    # ---- launches --------------------------------------------------------------------------------------------
    def _splitk(self, d: AmmcConvDesc, dev) -> None:
        """hand the call the split-K workspace (the library uses it only where a layer cannot fill the chip)"""
        wk = self._cur_ws["splitk"]
        d.splitk_ws, d.splitk_ws_floats = _ptr(wk), wk.numel()

    def _conv(self, x: Act, pk: dict, y: Act, stride=1, act=ACT_LRELU, n_store=0, y_f32=False):
        """3x3 conv of `x` (one Act, or the list of parts of a concatenation) into y (`y_f32`: an fp32 output from S16
        operands - the flow heads)"""
        parts = x if isinstance(x, (list, tuple)) else [x]
        s = torch.cuda.current_stream(y.buf.device).cuda_stream
        for i, (xp, (wp, cp)) in enumerate(zip(parts, pk["w"])):
            d = AmmcConvDesc()
            d.x, d.w, d.y = xp.tap0(), _ptr(wp), y.pix0()
            d.shift = _ptr(pk["bias"]) if i == 0 else None
            d.res = y.pix0() if i > 0 else None
            d.batch, d.height, d.width = y.B, y.H, y.W
            d.cin, d.ntaps, d.n, d.up, d.cgroup, d.x_step = cp, 9, pk["n"], 1, pk["n"], stride
            d.act = act if len(parts) == 1 else ACT_NONE
            d.n_store = n_store if n_store else (pk["co"] if pk["co"] != pk["n"] else 0)
            d.x_bs, d.x_rs, d.x_ps = xp.strides
            d.y_bs, d.y_rs, d.y_ps = y.strides
            d.r_bs, d.r_rs, d.r_ps = y.strides
            if self.s16:
                d.y_f32 = 1 if y_f32 else 0
                d.overflow_flag = self._cur_ws["flag"].data_ptr()
                self._splitk(d, y.buf.device)
                _chk(self.lib.ammc_conv_gemm_s16(C.byref(d), s), "flownet.conv(s16)")
            else:
                _chk(self.lib.ammc_conv_gemm_f32(C.byref(d), s), "flownet.conv")
        assert len(parts) == 1 or act == ACT_NONE, "a multi-part conv carries no activation in this network"

    def _deconv(self, x, pk: dict, y: Act, lrelu: bool):
        """ConvTranspose2d(k 4, s 2, p 1) of x (Act or parts) into y (twice the resolution) [+ LeakyReLU]"""
        parts = x if isinstance(x, (list, tuple)) else [x]
        s = torch.cuda.current_stream(y.buf.device).cuda_stream
        for i, (xp, (wp, cp, per)) in enumerate(zip(parts, pk["w"])):
            for ph in range(4):
                py, px = ph >> 1, ph & 1
                d = AmmcConvDesc()
                # window origin: input pixel ((py + 1) >> 1) - 1 + q' for output row 2 q' + py
                d.x = xp.pix0() + 4 * ((((py + 1) >> 1) - 1) * xp.rs + (((px + 1) >> 1) - 1) * xp.ps)
                d.w = _ptr(wp, ph * per)
                d.y = y.pix0() + 4 * (py * y.rs + px * y.ps)
                d.shift = _ptr(pk["bias"]) if i == 0 else None
                d.res = d.y if i > 0 else None
                d.batch, d.height, d.width = xp.B, xp.H, xp.W
                d.cin, d.ntaps, d.n, d.up, d.cgroup, d.x_step = cp, 4, pk["n"], 1, pk["n"], 1
                d.act = ACT_LRELU if (lrelu and len(parts) == 1) else ACT_NONE
                d.n_store = pk["co"] if pk["co"] != pk["n"] else 0
                d.x_bs, d.x_rs, d.x_ps = xp.strides
                d.y_bs, d.y_rs, d.y_ps = y.bs, 2 * y.rs, 2 * y.ps
                d.r_bs, d.r_rs, d.r_ps = y.bs, 2 * y.rs, 2 * y.ps
                if self.s16 and not pk.get("head"):
                    d.overflow_flag = self._cur_ws["flag"].data_ptr()
                    self._splitk(d, y.buf.device)
                    _chk(self.lib.ammc_conv_gemm_s16(C.byref(d), s), "flownet.deconv(s16)")
                else:
                    _chk(self.lib.ammc_conv_gemm_f32(C.byref(d), s), "flownet.deconv")
        if lrelu and len(parts) > 1:
            fn = self.lib.ammc_lrelu_s16 if self.s16 else self.lib.ammc_lrelu_f32
            _chk(fn(y.pix0(), *y.strides, y.B, y.H, y.W, y.c, SLOPE, s), "flownet.lrelu")

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        if not inputs.is_cuda:
            raise _lib.AmmcHipError("FlowNet2SD runs on the HIP kernels only: input must be a GPU tensor")
        if inputs.dim() != 5 or inputs.shape[1] != 3 or inputs.shape[2] != 2:
            raise ValueError("expected frame pairs [B, 3, 2, H, W]")
        B, _, _, H, W = inputs.shape
        if H % 64 or W % 64:
            raise ValueError(f"frame size {H}x{W} must be divisible by 64 (six stride-2 levels)")
        x = inputs.detach().float().contiguous()
        dev = x.device
        self._ensure_packs()
        ws, pk, lib = self._workspace(B, H, W, dev), self.packs, self.lib
        self._cur_ws = ws
        if self.s16:
            ws["flag"].zero_()
        s = torch.cuda.current_stream(dev).cuda_stream
        x0 = ws["x0"]
        if "prep" not in ws:
            ws["prep"] = torch.empty(lib.ammc_flownet_prep_scratch_doubles(B), device=dev, dtype=torch.float64)
        _chk(lib.ammc_flownet_prep_f32(_ptr(x), B, H, W, x0.pix0(), *x0.strides, float(self.m.rgb_max),
                                       ws["prep"].data_ptr(), s), "prep")
        cat2, cat3, cat4, cat5 = ws["cat2"], ws["cat3"], ws["cat4"], ws["cat5"]
        c2, d2 = cat2.slice(0, 128), cat2.slice(128, 64)
        c3, d3 = cat3.slice(0, 256), cat3.slice(256, 128)
        c4, d4 = cat4.slice(0, 512), cat4.slice(512, 256)
        c5, d5 = cat5.slice(0, 512), cat5.slice(512, 512)
        if self.s16:
            u3, u4, u5, u6 = ws["u3"], ws["u4"], ws["u5"], ws["u6"]
            x0s = ws["x0s"]
            _chk(lib.ammc_split_rows_f32(_ptr(x0.buf), x0.buf.numel(), _ptr(x0s.buf), s), "split_rows(x0)")
            x0 = x0s
        else:
            u3, u4, u5, u6 = cat2.slice(192, 4), cat3.slice(384, 4), cat4.slice(768, 4), cat5.slice(1024, 4)

        def upflow(f: Act, name: str, u: Act):
            """the 2 -> 2 transposed conv of a flow estimate (fp32 kernels); S16 form: into the fp32 staging buffer, then
            re-encoded as the 8-channel S16 part of the concatenation"""
            if not self.s16:
                self._deconv(f, pk[name], u, lrelu=False)
                return
            uf = ws["uf" + name[2]]
            self._deconv(f, pk[name], uf, lrelu=False)
            _chk(lib.ammc_split_rows_f32(_ptr(uf.buf), uf.buf.numel(), _ptr(u.buf), s), "split_rows(flow)")

        self._conv(x0, pk["conv0"], ws["c0"])
        self._conv(ws["c0"], pk["conv1"], ws["t1"], stride=2)
        self._conv(ws["t1"], pk["conv1_1"], ws["c1"])
        self._conv(ws["c1"], pk["conv2"], ws["t2"], stride=2)
        self._conv(ws["t2"], pk["conv2_1"], c2)
        self._conv(c2, pk["conv3"], ws["t3"], stride=2)
        self._conv(ws["t3"], pk["conv3_1"], c3)
        self._conv(c3, pk["conv4"], ws["t4"], stride=2)
        self._conv(ws["t4"], pk["conv4_1"], c4)
        self._conv(c4, pk["conv5"], ws["t5"], stride=2)
        self._conv(ws["t5"], pk["conv5_1"], c5)
        self._conv(c5, pk["conv6"], ws["t6"], stride=2)
        self._conv(ws["t6"], pk["conv6_1"], ws["c6"])
        # decoder (models.py:31-54)
        self._conv(ws["c6"], pk["predict_flow6"], ws["f6"], act=ACT_NONE, y_f32=True)
        upflow(ws["f6"], "up6", u6)
        self._deconv(ws["c6"], pk["deconv5"], d5, lrelu=True)
        p5 = [c5, d5, u6]
        self._conv(p5, pk["inter_conv5"], ws["ic5"], act=ACT_NONE)
        self._conv(ws["ic5"], pk["predict_flow5"], ws["f5"], act=ACT_NONE, y_f32=True)
        upflow(ws["f5"], "up5", u5)
        self._deconv(p5, pk["deconv4"], d4, lrelu=True)
        p4 = [c4, d4, u5]
        self._conv(p4, pk["inter_conv4"], ws["ic4"], act=ACT_NONE)
        self._conv(ws["ic4"], pk["predict_flow4"], ws["f4"], act=ACT_NONE, y_f32=True)
        upflow(ws["f4"], "up4", u4)
        self._deconv(p4, pk["deconv3"], d3, lrelu=True)
        p3 = [c3, d3, u4]
        self._conv(p3, pk["inter_conv3"], ws["ic3"], act=ACT_NONE)
        self._conv(ws["ic3"], pk["predict_flow3"], ws["f3"], act=ACT_NONE, y_f32=True)
        upflow(ws["f3"], "up3", u3)
        self._deconv(p3, pk["deconv2"], d2, lrelu=True)
        p2 = [c2, d2, u3]
        self._conv(p2, pk["inter_conv2"], ws["ic2"], act=ACT_NONE)
        f2 = ws["f2"]
        # the last S16 activation is written: hand the flag to the host now (4 bytes to pinned memory + an event), two
        # small launches before the forward ends
        self.flag_event = None
        if self.s16:
            if "flag_host" not in ws:
                ws["flag_host"], ws["flag_ev"] = torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event()
            ws["flag_host"].copy_(ws["flag"], non_blocking=True)
            ws["flag_ev"].record()
            self.flag_event, self.flag_host, self.flag_dev = ws["flag_ev"], ws["flag_host"], ws["flag"]
        self._conv(ws["ic2"], pk["predict_flow2"], f2, act=ACT_NONE, y_f32=True)
        out = torch.empty(B, 2, H, W, device=dev, dtype=torch.float32)
        _chk(lib.ammc_upsample4_bilinear_f32(f2.pix0(), *f2.strides, B, f2.H, f2.W, 2, float(self.m.div_flow), _ptr(out), s),
             "upsample")
        return out


def _conv(cin, cout, stride=1):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=True), nn.LeakyReLU(SLOPE, inplace=True))


def _deconv(cin, cout):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True), nn.LeakyReLU(SLOPE, inplace=True))


def _iconv(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=True))


class FlowNet2SD(nn.Module):
    """parameter holders with the reference's names and shapes; `forward` (eval mode) runs on the HIP kernels"""

    def __init__(self, batchNorm: bool = False, div_flow: float = 20):
        super().__init__()
        if batchNorm:
            raise NotImplementedError("FlowNet2SD(batchNorm=True) is not built; the reference constructs it with "
                                      "batchNorm=False (models/__init__.py:126, flownet2/models.py:10)")
        self.batchNorm, self.rgb_max, self.div_flow = False, 255.0, div_flow
        self.precision = DEFAULT_PRECISION
        self.conv0 = _conv(6, 64)
        self.conv1, self.conv1_1 = _conv(64, 64, 2), _conv(64, 128)
        self.conv2, self.conv2_1 = _conv(128, 128, 2), _conv(128, 128)
        self.conv3, self.conv3_1 = _conv(128, 256, 2), _conv(256, 256)
        self.conv4, self.conv4_1 = _conv(256, 512, 2), _conv(512, 512)
        self.conv5, self.conv5_1 = _conv(512, 512, 2), _conv(512, 512)
        self.conv6, self.conv6_1 = _conv(512, 1024, 2), _conv(1024, 1024)
        self.deconv5, self.deconv4 = _deconv(1024, 512), _deconv(1026, 256)
        self.deconv3, self.deconv2 = _deconv(770, 128), _deconv(386, 64)
        self.inter_conv5, self.inter_conv4 = _iconv(1026, 512), _iconv(770, 256)
        self.inter_conv3, self.inter_conv2 = _iconv(386, 128), _iconv(194, 64)
        self.predict_flow6, self.predict_flow5 = nn.Conv2d(1024, 2, 3, 1, 1), nn.Conv2d(512, 2, 3, 1, 1)
        self.predict_flow4, self.predict_flow3 = nn.Conv2d(256, 2, 3, 1, 1), nn.Conv2d(128, 2, 3, 1, 1)
        self.predict_flow2 = nn.Conv2d(64, 2, 3, 1, 1)
        self.upsampled_flow6_to_5 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow5_to_4 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow4_to_3 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow3_to_2 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        object.__setattr__(self, "_engine", None)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise NotImplementedError("FlowNet2SD is a frozen estimator here: call .eval() (train_helper.py:284)")
        if self._engine is None or self._engine.s16 != (self.precision == "s16"):
            object.__setattr__(self, "_engine", _Engine(self, self.precision))
        eng = self._engine
        with torch.no_grad():
            out = eng.forward(inputs)
            # S16 range guard (as the generator's, unet.py): the split-fp16 kernels carry activations as (hi, lo) halves,
            # one beyond 65504 becomes inf and the flows NaN - silently, in a `no_grad` flow extraction.  `s16_guard`:
            # True (default) = read the sticky device flag once per forward and recompute the batch on the exact-fp32
            # kernels when it is set; "defer" = leave it in `last_overflow` (a [1] int32 tensor ON THE DEVICE) for a
            # trainer that folds it into its own verdict (harness.train_step_gan) - no host wait; False = off.
            guard = getattr(self, "s16_guard", True)
            object.__setattr__(self, "last_overflow", None)
            if eng.s16 and eng.flag_event is not None and guard:
                if guard == "defer":
                    object.__setattr__(self, "last_overflow", eng.flag_dev.clone())
                else:
                    eng.flag_event.synchronize()
                    if int(eng.flag_host[0]) != 0:
                        if getattr(self, "_engine_fp32", None) is None:
                            object.__setattr__(self, "_engine_fp32", _Engine(self, "fp32"))
                        out = self._engine_fp32.forward(inputs)
                        object.__setattr__(self, "s16_fallbacks", getattr(self, "s16_fallbacks", 0) + 1)
            return out
