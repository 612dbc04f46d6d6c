"""`ammc_conv_wgrad_s16` (csrc/wgrad_s16.hip: fp16 MFMA, transposed LDS fragment reads) against an fp64 weight gradient
of the same operands: dW[n][c][r][s] = sum_m G[m][n] * A[m + (r, s)][c].  Includes a gradient-sized G (1e-7) that goes
through the device-side power-of-two rescaling, a first-layer shape (16 input channels, K padding) and a pixel count
that is not a multiple of the 32-pixel chunk.  Tolerance 3e-6 of max|ref| (fp32-equivalent operands, fp32 atomics)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from ammcnet_aaai2021_amd import _lib, synthetic as S
from ammcnet_aaai2021_amd._lib import AmmcWgradDesc
from ammcnet_aaai2021_amd.engine import Act, _ptr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# W % 32 == 0, H even, n % 64 == 0, cin % 64 == 0 go to the three-MFMA halo-patch kernel (csrc/wgrad_tap3_s16.hip);
# W % 32 == 0 and H % 4 == 0 with (n % 128, cin % 16) or (n % 64, cin % 32) to the four-product one
# (csrc/wgrad_tap_s16.hip), the rest to the im2col one (csrc/wgrad_s16.hip); AMMC_WGRAD_TAP=2 / =0 force the latter two
@pytest.mark.parametrize("B,H,W,cin,n,gmag", [(2, 32, 32, 64, 128, 1.0), (3, 20, 24, 128, 64, 3e-7), (2, 16, 16, 16, 64, 1e-6),
                                               (1, 9, 7, 32, 32, 1.0), (4, 64, 64, 64, 64, 2e-8), (1, 8, 64, 16, 128, 1e-5),
                                               (2, 16, 32, 128, 64, 1.0), (3, 12, 96, 256, 256, 1e-3),
                                               # more shapes of the three-MFMA kernel: one patch row pair, a tiny gradient, five samples
                                               (1, 2, 32, 64, 128, 1.0), (2, 6, 64, 128, 256, 1e-6), (5, 10, 32, 512, 128, 3e-4),
                                               # its 4-row-patch forms: 32 gradient channels (output layer), 8 / 16 / 32 input
                                               # channels (first layers; the missing channels come from the zero buffer)
                                               (2, 16, 32, 64, 32, 1e-5), (2, 8, 64, 16, 64, 1.0), (1, 4, 32, 8, 128, 1e-3),
                                               (3, 12, 32, 32, 64, 1.0), (1, 20, 96, 128, 32, 1.0),
                                               # rolling halo: one-patch columns (every patch starts a column), runs that end mid-column
                                               (2, 2, 64, 64, 64, 1.0), (3, 2, 96, 128, 128, 1e-4), (1, 22, 32, 64, 64, 1.0)])
def test_wgrad_s16_vs_fp64(B, H, W, cin, n, gmag):
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tag = f"wg-{B}-{H}-{W}-{cin}-{n}"
    a = S.hashed_uniform(tag + "a", (B, H, W, cin)).to(DEV)
    g = (S.hashed_uniform(tag + "g", (B, H, W, n)) * gmag).to(DEV)
    A32 = Act(torch.zeros(B, H + 2, W + 2, cin, device=DEV), B, H, W, cin, 0, 1)
    A32.interior().copy_(a)
    G32 = Act(torch.zeros(B, H + 2, W + 2, n, device=DEV), B, H, W, n, 0, 1)
    G32.interior().copy_(g)
    A16 = Act(torch.empty_like(A32.buf), B, H, W, cin, 0, 1)
    G16 = Act(torch.empty_like(G32.buf), B, H, W, n, 0, 1)
    _lib.check(lib.ammc_split_rows_f32(_ptr(A32.buf), A32.buf.numel(), _ptr(A16.buf), s), "split a")
    amax = torch.zeros(256, dtype=torch.int32, device=DEV)
    inv = torch.empty(8, device=DEV)
    _lib.check(lib.ammc_absmax_bits_f32(_ptr(G32.buf), G32.buf.numel(), amax.data_ptr(), s), "absmax")
    _lib.check(lib.ammc_split_rows_scaled_f32(_ptr(G32.buf), G32.buf.numel(), _ptr(G16.buf), amax.data_ptr(), _ptr(inv),
                                              8, s), "split g")
    factor = 1.0 / float(inv[0])
    assert 1024.0 <= float(g.abs().max()) * factor < 2048.0                   # the maximum lands in [2^10, 2^11)
    kpad = (9 * cin + 31) // 32 * 32
    dwp = torch.zeros(max(n, 32), kpad, device=DEV)
    zeros = torch.zeros(1024, device=DEV)
    d = AmmcWgradDesc()
    d.g, d.a, d.dw, d.zeros = G16.pix0(), A16.tap0(), _ptr(dwp), _ptr(zeros)
    d.batch, d.height, d.width, d.n, d.cin, d.ntaps, d.a_step = B, H, W, n, cin, 9, 1
    d.g_bs, d.g_rs, d.g_ps = G16.strides
    d.a_bs, d.a_rs, d.a_ps = A16.strides
    _lib.check(lib.ammc_conv_wgrad_s16(C.byref(d), _ptr(inv), s), "wgrad_s16")
    dw = torch.empty(n, cin, 3, 3, device=DEV)
    _lib.check(lib.ammc_unpack_conv_wgrad_f32(_ptr(dwp), n, cin, 3, cin, _ptr(dw), s), "unpack")
    # fp64 reference: the weight gradient of conv2d(a, w) against the upstream gradient g
    a64 = a.double().cpu().permute(0, 3, 1, 2).contiguous()
    g64 = g.double().cpu().permute(0, 3, 1, 2).contiguous()
    w = torch.zeros(n, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(a64, w, padding=1) * g64).sum().backward()
    err = float((dw.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
    # AMMC_WGRAD_G11=1 (opt-in; the child run below): the rolling-halo instances take the GRADIENT operand with its 11-bit
    # hi half, two MFMAs per product block (wgrad_tap3_s16.hip) - a relative 2^-12 rounding of every g, ~1e-4 of max |dw|
    # on these random operands; the default keeps all three products: 3e-6
    import os
    g11 = (os.environ.get("AMMC_WGRAD_G11", "0") != "0" and os.environ.get("AMMC_WGRAD_ROLL", "1") != "0"
           and cin % 64 == 0 and n % 64 == 0 and W % 32 == 0 and H % 2 == 0)
    tol = 3e-4 if g11 else 3e-6
    assert err <= tol, (err, tol)
    if g11:
        assert err >= 3e-6, err                       # (the switch did switch)
    # the slab form of the same launch (round 5): split partials stored as slabs and summed straight into the OIHW gradient
    # - no atomics, no unpack; available exactly where the three-MFMA halo-patch kernel takes the shape.  Same 3e-6 against
    # fp64, equal to the atomics form to summation order, and bit-identical between two launches (fixed summation order)
    need = int(lib.ammc_conv_wgrad_s16_slab_floats(C.byref(d)))
    takes = (W % 32 == 0 and H % 2 == 0 and ((cin % 64 == 0 and (n % 64 == 0 or (n == 32 and H % 4 == 0))) or
                                             (cin <= 32 and n % 64 == 0 and H % 4 == 0)))
    assert (need > 0) == takes, (need, takes)
    if need > 0:
        assert need % (n * kpad) == 0
        outs = []
        for _ in range(2):
            slabs = torch.full((need + 64,), float("nan"), device=DEV)           # (every element that is read must have been written)
            dws = torch.empty(n, cin, 3, 3, device=DEV)
            _lib.check(lib.ammc_conv_wgrad_s16_slabs(C.byref(d), _ptr(inv), _ptr(slabs), need, _ptr(dws), n, cin, s), "wgrad slabs")
            outs.append(dws)
        assert torch.equal(outs[0], outs[1])
        err_s = float((outs[0].double().cpu() - w.grad).abs().max() / w.grad.abs().max())
        assert err_s <= tol, (err_s, tol)
        assert float((outs[0] - dw).abs().max()) <= 2e-6 * float(dw.abs().max())
        assert lib.ammc_conv_wgrad_s16_slabs(C.byref(d), _ptr(inv), _ptr(slabs), need - 1, _ptr(dws), n, cin, s) == -1   # AMMC_EINVAL


def test_two_product_form_of_the_gradient_operand():
    """`AMMC_WGRAD_G11=1` (opt-in): the gradient operand with its hi half only, two MFMAs per product block - the same
    launches within 3e-4 of fp64 (and measurably NOT at 3e-6: the switch did switch), in a child process (the switch is
    read once per process)"""
    import os
    import subprocess
    import sys
    if "AMMC_WGRAD_G11" in os.environ or "AMMC_WGRAD_ROLL" in os.environ:
        pytest.skip("already inside a run with a switch set")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "test_wgrad_s16_vs_fp64"],
                       env=dict(os.environ, AMMC_WGRAD_G11="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_per_patch_halo_form_still_passes():
    """`AMMC_WGRAD_ROLL=0` (the A/B switch of the rolling input halo, read once per process): the per-patch halo form of the
    same kernels against the same fp64 references, in a child process"""
    import os
    import subprocess
    import sys
    if "AMMC_WGRAD_ROLL" in os.environ or "AMMC_WGRAD_G11" in os.environ:
        pytest.skip("already inside a run with a switch set")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "test_wgrad_s16_vs_fp64"],
                       env=dict(os.environ, AMMC_WGRAD_ROLL="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
