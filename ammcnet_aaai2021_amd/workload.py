"""Workload definitions of BASELINE.json's configs that are not a model call: the algorithmic-work formula the
roofline figures divide by, and config 5 (memory addressing alone: 8192 slots x 512-d, fp16 MFMA operands, rows
sharded over ranks).

`MemoryStress` is the N-sharded entry of SURVEY.md 8(e): the feature rows x [N, D] are independent, so rank r of R
takes rows [r N / R, (r + 1) N / R) (whole 1024-row frames) and a replica of the codebook (8 MB fp16); there is no
exchange step - the only collective is the final sum of the commit partials, outside the timed region.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib
from .engine import _Packer, _ptr


def fwd_flops_per_clip(h: int = 256, w: int = 256, in_channel=(12, 6), out_channel=(3, 2),
                       embed_dim: int = 64, n_embed: int = 256, k: int = 2) -> float:
    """Algorithmic forward FLOPs (2*MAC) of one dual-stream clip through `twostream.forward`
    (reference models/unet.py:981-1007): 3x3 convs, ConvTranspose, 1x1, distance GEMM - BN / ReLU / pool / tanh not
    counted (168.10 GFLOP at the shipped config, 256 slots)."""
    def dc(cin, cout, hh, ww):
        return 2.0 * hh * ww * 9 * (cin * cout + cout * cout)
    total = 0.0
    for cin, cout in zip(in_channel, out_channel):
        total += dc(cin, 64, h, w) + dc(64, 128, h // 2, w // 2) + dc(128, 256, h // 4, w // 4)
        total += dc(256, 512, h // 8, w // 8)
        n = (h // 8) * (w // 8)
        total += 2.0 * n * (512 * embed_dim + embed_dim * n_embed + k * embed_dim * 512)
        for c, s in ((512, 4), (256, 2), (128, 1)):
            hh, ww = h // s, w // s                     # output resolution of this up block
            total += 2.0 * (hh // 2) * (ww // 2) * c * (c // 2) * 4     # ConvT
            total += dc(c, c // 2, hh, ww)
        total += 2.0 * h * w * 9 * 64 * cout
    total += 2 * dc(512, 512, h // 8, w // 8)
    return total


def shard_rows(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) frame (1024 feature rows each) of `rank`: contiguous, sizes differ by at most one frame"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n_frames, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class MemoryStress:
    """BASELINE.json configs[4]: `Quantize_topk.forward` (unet.py:282-297, 310-313) alone at D=512, M=8192, k=2 on the
    fp16-operand kernel (`ammc_memory_topk_fwd_f16r`, or `ammc_memory_topk_fwd_f16` - see `rows_in_registers`); the codebook is
    packed once, every `run` is ONE launch on the rows this rank owns."""

    def __init__(self, embed: torch.Tensor, k: int = 2, rows_in_registers=None):
        """`rows_in_registers`: True = csrc/memory_topk_f16r.hip (round 5: feature rows resident in registers, codebook
        tiles through LDS), False = csrc/memory_topk_f16.hip (rows in LDS, codebook L2 -> registers); None = the package
        default (`ops.F16_ROWS_IN_REGISTERS`, AMMC_MEMORY_F16_FORM)"""
        if not embed.is_cuda:
            raise _lib.AmmcHipError("the HIP path needs tensors on the GPU; there is no CPU fallback")
        from . import ops
        self.lib = _lib.load()
        self.d, self.m = embed.shape
        self.k = k
        self.dev = embed.device
        self.embed = embed.detach().float().contiguous()
        self.rows_in_registers = ops.F16_ROWS_IN_REGISTERS if rows_in_registers is None else bool(rows_in_registers)
        self.kernel = "memory_topk_f16r" if self.rows_in_registers else "memory_topk_f16"
        s = torch.cuda.current_stream(self.dev).cuda_stream
        self.e_md, _ = _Packer(self.dev).codebook(self.embed)
        if self.rows_in_registers:
            self.tiles = torch.empty(self.lib.ammc_codebook_f16_tiles_bytes(self.d, self.m), device=self.dev, dtype=torch.uint8)
            _lib.check(self.lib.ammc_pack_codebook_f16_tiles(_ptr(self.embed), self.d, self.m, self.tiles.data_ptr(), s),
                       "pack_codebook_f16_tiles")
        else:
            mpad = (self.m + 31) // 32 * 32
            self.e_kblk = torch.empty((self.d // 8, mpad, 8), device=self.dev, dtype=torch.float16)
            self.enorm = torch.empty(self.m, device=self.dev, dtype=torch.float32)
            _lib.check(self.lib.ammc_pack_codebook_f16(_ptr(self.embed), self.d, self.m, self.e_kblk.data_ptr(),
                                                       _ptr(self.enorm), s), "pack_codebook_f16")
        self._out = {}

    def flops(self, n_rows: int) -> float:
        return 2.0 * n_rows * self.d * self.m

    def algorithmic_bytes(self, n_rows: int) -> float:
        """x read (fp32) + codebook (fp16) + gathered rows written (fp32) + indices"""
        return 4.0 * n_rows * self.d + 2.0 * self.d * self.m + 4.0 * n_rows * self.k * self.d + 4.0 * n_rows * self.k

    def run(self, x: torch.Tensor):
        """x [n, D] fp32 on the device -> (q_topk [n, k D], commit partial sums, q_one [n, D], idx [n, k]); output
        buffers are reused across calls of the same n"""
        n = x.shape[0]
        o = self._out.get(n)
        if o is None:
            o = self._out[n] = dict(
                idx=torch.empty((n, self.k), device=self.dev, dtype=torch.int32),
                qk=torch.empty((n, self.k * self.d), device=self.dev, dtype=torch.float32),
                q1=torch.empty((n, self.d), device=self.dev, dtype=torch.float32),
                part=torch.empty(self.lib.ammc_memory_topk_f16r_blocks(n) if self.rows_in_registers else
                                 self.lib.ammc_memory_topk_f16_blocks(n), device=self.dev, dtype=torch.float32))
        s = torch.cuda.current_stream(self.dev).cuda_stream
        if self.rows_in_registers:
            _lib.check(self.lib.ammc_memory_topk_fwd_f16r(_ptr(x), self.tiles.data_ptr(), _ptr(self.e_md), n, self.d, self.m,
                                                          self.k, o["idx"].data_ptr(), _ptr(o["qk"]), _ptr(o["q1"]),
                                                          _ptr(o["part"]), s), "memory_topk_f16r")
        else:
            _lib.check(self.lib.ammc_memory_topk_fwd_f16(_ptr(x), self.e_kblk.data_ptr(), _ptr(self.e_md), _ptr(self.enorm),
                                                         n, self.d, self.m, self.k, o["idx"].data_ptr(), _ptr(o["qk"]),
                                                         _ptr(o["q1"]), _ptr(o["part"]), s), "memory_topk_f16")
        return o["qk"], o["part"], o["q1"], o["idx"]
