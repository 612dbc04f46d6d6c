"""Host-side mirror of the reference's model interface for the hot path.

Same class names, constructor arguments, attribute names, `forward` signatures,
return values and `state_dict` layout as the live classes of the reference's
`Code/models/unet.py` (double_conv :8-20, inconv :23-30, down :33-41, up :44-59,
UNet :61-83, Quantize_topk :267-316, enc_quan_dec_topk :318-331,
enc_quan_dec_res_topk :379-387, UNetMem_v7 :908-937, bridge :956-965,
twostream :967-1007, get_unet :1130, get_unet_vq_topk_res :1213, get_twostream
:1241), so a checkpoint of the reference loads with `load_state_dict(strict=True)`
and `Code/main/run_test` / `run_train` can use these classes unchanged.

What differs is everything underneath: the torch.nn leaf modules here only
HOLD parameters.  `forward` never calls ATen convolution / batch-norm / topk:
it hands raw device pointers to the gfx950 kernels of libammc_hip.so through
the launch plans in `engine.py`.  If the library is missing, or the tensors are
not on the GPU, forward raises; there is no fallback.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from .engine import EvalEngine
from .train import BlockEngine, BlockFunction, HipPathFunction, MemoryBlockEngine, MemoryBlockFunction, TrainEngine


# Inference arithmetic: "s16" (default) = split-fp16 MFMA with fp32 accumulation: 22 significant bits per operand,
# as accurate against fp64 as native fp32, at 3x the speed of "fp32" = exact fp32 MFMA
# (v_mfma_f32_32x32x2_f32).  Per model: `model.precision = "..."`; process-wide: AMMC_PRECISION.
# The one thing S16 does not share with fp32 is the RANGE of the hi half (|v| <= 65504): every S16 epilogue raises a
# device flag beyond it and the guard below (on by default, `model.s16_guard = False` turns it off) recomputes the
# batch on the exact-fp32 kernels, so the default mode never returns frames computed from a saturated activation.
DEFAULT_PRECISION = os.environ.get("AMMC_PRECISION", "s16")
DEFAULT_S16_GUARD = os.environ.get("AMMC_S16_GUARD", "1") != "0"      # (A/B measurements of the guard's cost only)


def _run_memory(mod, kind: str, x):
    """training-mode forward of the memory block on its own (`Quantize_topk`, `enc_quan_dec_topk`,
    `enc_quan_dec_res_topk`): one autograd node over the HIP kernels; the EMA buffers are updated in place as the
    reference does whenever `self.training` (models/unet.py:298-309)"""
    eng = mod.__dict__.get("_block_engine")
    if eng is None or eng.kind != kind:
        eng = MemoryBlockEngine(mod, kind)
        object.__setattr__(mod, "_block_engine", eng)
    return MemoryBlockFunction.apply(eng, x, *mod.parameters())


def _run_block(mod, kind: str, owner, *inputs):
    """training-mode forward of a stand-alone block: one autograd node over the HIP kernels (train.BlockEngine)"""
    eng = owner.__dict__.get("_block_engine")
    tprec = getattr(owner, "train_precision", None)
    if eng is None or (tprec is not None and eng.precision != tprec):
        eng = BlockEngine(mod, kind, tprec)
        object.__setattr__(owner, "_block_engine", eng)
    return BlockFunction.apply(eng, len(inputs), *inputs, *mod.parameters())


def _fp32_engine(mod, kind: str) -> EvalEngine:
    """the exact-fp32 plans of a model whose default engine is S16 (built on first use: the overflow fallback)"""
    if getattr(mod, "_engine_fp32", None) is None:
        object.__setattr__(mod, "_engine_fp32", EvalEngine(mod, kind, "fp32"))
    return mod._engine_fp32


def _run(mod, kind: str, n_inputs: int, *inputs):
    """dispatch a model-boundary forward to the eval plan or to the training Function"""
    if not mod.training:
        prec = getattr(mod, "precision", None) or DEFAULT_PRECISION
        if mod._engine is None or mod._engine.precision != prec:
            object.__setattr__(mod, "_engine", EvalEngine(mod, kind, prec))
        out = mod._engine.forward(*inputs)
        object.__setattr__(mod, "_last_engine", mod._engine)          # the engine that produced `out` (quant_befor / quant_after)
        if prec == "s16" and getattr(mod, "s16_guard", DEFAULT_S16_GUARD) and not torch.cuda.is_current_stream_capturing():
            # Reading the flag is one 4-byte copy to pinned memory queued ahead of the two output layers
            # (EvalEngine._launch_all): the wait ends while the device still has work queued, < 1 % at batch 16
            # (DESIGN.md section 3); the harness loop avoids even that, see `forward_scored(defer_guard=True)`.
            # Under a caller's stream capture nothing may be read back or waited for: the flag stays sticky on the
            # device and the caller checks it after the replay (`model._engine.take_overflow()` / `.overflowed()`).
            if mod._engine.overflowed():
                out = _fp32_engine(mod, kind).forward(*inputs)
                object.__setattr__(mod, "_last_engine", mod._engine_fp32)
                object.__setattr__(mod, "s16_fallbacks", getattr(mod, "s16_fallbacks", 0) + 1)
        return out
    tprec = getattr(mod, "train_precision", None)            # None: train.TRAIN_PRECISION (AMMC_TRAIN_PRECISION, "s16")
    if mod._train_engine is None or (tprec is not None and mod._train_engine.precision != tprec):
        object.__setattr__(mod, "_train_engine", TrainEngine(mod, kind, tprec))
    return HipPathFunction.apply(mod._train_engine, n_inputs, *inputs, *mod.parameters())


class double_conv(nn.Module):
    """[conv3x3 (pad 1, no bias) -> BatchNorm2d -> ReLU] x 2; BN + ReLU live in the conv epilogue."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, padding=1, bias=False),
                                  nn.BatchNorm2d(out_ch),
                                  nn.ReLU(inplace=True),
                                  nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False),
                                  nn.BatchNorm2d(out_ch),
                                  nn.ReLU(inplace=True))

    def forward(self, x):
        if self.training:
            return _run_block(self, "double_conv", self, x)
        return ops.double_conv_eval(self, x)


class inconv(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = double_conv(in_ch, out_ch)

    def forward(self, x):
        return self.conv(x)


class down(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.mpconv = nn.Sequential(nn.MaxPool2d(2), double_conv(in_ch, out_ch))

    def forward(self, x):
        if self.training:
            return _run_block(self, "down", self, x)
        return ops.double_conv_eval(self.mpconv[1], x, pool_first=True)


class up(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_ch, in_ch // 2, 2, stride=2)
        self.conv = double_conv(in_ch, out_ch)

    def forward(self, x1, x2):
        if self.training:
            return _run_block(self, "up", self, x1, x2)
        return ops.up_eval(self, x1, x2)


class UNet(nn.Module):
    def __init__(self, input_channels, output_channel=3):
        super().__init__()
        self.inc = inconv(input_channels, 64)
        self.down1 = down(64, 128)
        self.down2 = down(128, 256)
        self.down3 = down(256, 512)
        self.up1 = up(512, 256)
        self.up2 = up(256, 128)
        self.up3 = up(128, 64)
        self.outc = nn.Conv2d(64, output_channel, kernel_size=3, padding=1)
        self._engine = self._train_engine = None
        self._param_epoch = 0

    def forward(self, x):
        out = _run(self, "unet", 1, x)
        return out[0] if self.training else out


class Quantize_topk(nn.Module):
    """memory module: L2 distance to every slot, k nearest slots concatenated, commit distance"""

    def __init__(self, dim, n_embed, decay=0.99, eps=1e-5, k=1):
        super().__init__()
        self.dim = dim
        self.n_embed = n_embed
        self.decay = decay
        self.eps = eps
        self.k = k
        embed = torch.randn(dim, n_embed)
        self.register_buffer("embed", embed)
        self.register_buffer("cluster_size", torch.zeros(n_embed))
        self.register_buffer("embed_avg", embed.clone())

    def forward(self, input):
        if self.training:
            return _run_memory(self, "quantize", input)
        return ops.quantize_topk_eval(self, input)

    def embed_code(self, embed_id):
        return ops.embed_rows(self.embed, embed_id)


class enc_quan_dec_topk(nn.Module):
    def __init__(self, in_c, embed_dim, n_embed, k=1):
        super().__init__()
        self.enc = nn.Conv2d(in_c, embed_dim, 1)
        self.quantize = Quantize_topk(dim=embed_dim, n_embed=n_embed, k=k)
        self.dec = nn.Conv2d(embed_dim * k, in_c, 1)

    def forward(self, x):
        if self.training:
            return _run_memory(self, "vq", x)
        return ops.vq_block_eval(self, x, residual=False)


class enc_quan_dec_res_topk(nn.Module):
    def __init__(self, in_c, embed_dim, n_embed, k=1):
        super().__init__()
        self.quan = enc_quan_dec_topk(in_c, embed_dim, n_embed, k=k)

    def forward(self, x):
        if self.training:
            return _run_memory(self.quan, "vq_res", x)
        return ops.vq_block_eval(self.quan, x, residual=True)


class UNetMem_v7(nn.Module):
    def __init__(self, input_channels=3, output_channel=3, embed_dim=64, n_embed=512, k=1,
                 layer_nums=4, features_root=64):
        super().__init__()
        self.inc = inconv(input_channels, 64)
        self.down1 = down(64, 128)
        self.down2 = down(128, 256)
        self.down3 = down(256, 512)
        self.up1 = up(512, 256)
        self.up2 = up(256, 128)
        self.up3 = up(128, 64)
        self.outc = nn.Conv2d(64, output_channel, kernel_size=3, padding=1)
        self.vq_down3 = enc_quan_dec_res_topk(512, embed_dim, n_embed, k=k)
        self._engine = self._train_engine = None
        self._param_epoch = 0

    def forward(self, x):
        return _run(self, "unetmem", 1, x)


class bridge(nn.Module):
    """AMFT: x = zx + O2F(zy), y = zy + F20(zx); the adds are conv epilogues"""

    def __init__(self, in_c=64):
        super().__init__()
        self.O2F = double_conv(in_c, in_c)
        self.F20 = double_conv(in_c, in_c)

    def forward(self, zx, zy):
        if self.training:
            return _run_block(self, "bridge", self, zx, zy)
        return (ops.double_conv_eval(self.O2F, zy, residual=zx),
                ops.double_conv_eval(self.F20, zx, residual=zy))


class twostream(nn.Module):
    def __init__(self, rgb_in_c, rgb_out_c, op_in_c, op_out_c, embed_dim=64, n_embed=512, k=1,
                 layer_nums=4, features_root=64):
        super().__init__()
        self.rgb = UNetMem_v7(rgb_in_c, rgb_out_c, embed_dim, n_embed, k, layer_nums, features_root)
        self.op = UNetMem_v7(op_in_c, op_out_c, embed_dim, n_embed, k, layer_nums, features_root)
        self.bridge = bridge(in_c=512)
        self._engine = self._train_engine = None
        self._param_epoch = 0

    def forward(self, rgb_x, op_x):
        out = _run(self, "twostream", 2, rgb_x, op_x)
        if self.training:
            rgb, op, rd, od, rq, oq = out
            st = self._train_engine._last["streams"][0]
            befor, after = st.x4, st.x4q
            out = (rgb, op, (rd, od), (rq, oq))
            self._quant_src = (None, befor, after)
        else:
            eng = getattr(self, "_last_engine", None) or self._engine      # the fp32 engine after an S16 range fallback
            st = eng._last["streams"][0]
            self._quant_src = (eng, st.x4, st.x4q)
        return out

    def forward_scored(self, rgb_x, op_x, rgb_target, op_target=None, defer_guard: bool = False, exact: bool = False):
        """eval forward + per-sample PSNR of the predicted frames against `rgb_target` (and `op_target`), with the
        squared error accumulated inside the `outc` kernel (the scoring tail of
        run_helper/test_helper.py:445-454 without re-reading the frames).  Returns (forward's 4-tuple,
        rgb_psnr [B], op_psnr [B] or None).

        S16 range guard: by default as in `forward` (one flag read, fp32 recomputation on overflow).  With
        `defer_guard=True` nothing is read back: `self.last_overflow` is a [1] float tensor ON THE DEVICE (1 = an
        activation of this batch left the half range; the sticky flag is cleared by a queued kernel), so a harness
        can queue batches back to back, copy scores and flags to the host once, and re-run only the flagged batches
        with `exact=True` (the fp32 kernels) - `harness.evaluate_dataset` does exactly that."""
        if self.training:
            raise NotImplementedError("forward_scored is an evaluation entry: call .eval() first")
        prec = "fp32" if exact else (getattr(self, "precision", None) or DEFAULT_PRECISION)
        if exact and (getattr(self, "precision", None) or DEFAULT_PRECISION) != "fp32":
            eng = _fp32_engine(self, "twostream")
        else:
            if self._engine is None or self._engine.precision != prec:
                object.__setattr__(self, "_engine", EvalEngine(self, "twostream", prec))
            eng = self._engine
        out = eng.forward(rgb_x, op_x, targets=(rgb_target, op_target))
        self.last_overflow = None
        if prec == "s16" and getattr(self, "s16_guard", DEFAULT_S16_GUARD):
            if defer_guard:
                self.last_overflow = eng.take_overflow()
            elif eng.overflowed():
                eng = _fp32_engine(self, "twostream")
                out = eng.forward(rgb_x, op_x, targets=(rgb_target, op_target))
                object.__setattr__(self, "s16_fallbacks", getattr(self, "s16_fallbacks", 0) + 1)
        st = eng._last["streams"][0]
        self._quant_src = (eng, st.x4, st.x4q)
        return out, eng.last_psnr[0], eng.last_psnr[1]

    # reference side effects (unet.py:986, 988): `quant_befor` / `quant_after`, which nothing reads.
    # Served lazily as NCHW tensors of the workspace (valid until the next forward of that shape).
    @property
    def quant_befor(self):
        eng, a, _ = self._quant_src
        return eng.act_nchw(a) if eng is not None else a.interior().permute(0, 3, 1, 2)

    @property
    def quant_after(self):
        eng, _, a = self._quant_src
        return eng.act_nchw(a) if eng is not None else a.interior().permute(0, 3, 1, 2)


def get_unet(in_channel, out_channel, embed_dim=0, n_embed=0, k=0):
    return UNet(in_channel, out_channel)


def get_unet_vq_topk_res(in_channel, out_channel, embed_dim=64, n_embed=512, k=1):
    return UNetMem_v7(in_channel, out_channel, embed_dim, n_embed, k)


def get_twostream(in_channel, out_channel, embed_dim, n_embed, k, layer_nums=4, features_root=64):
    rgb_in_c, op_in_c = in_channel
    rgb_out_c, op_out_c = out_channel
    return twostream(rgb_in_c=rgb_in_c, rgb_out_c=rgb_out_c, op_in_c=op_in_c, op_out_c=op_out_c,
                     embed_dim=embed_dim, n_embed=n_embed, k=k, layer_nums=layer_nums,
                     features_root=features_root)
