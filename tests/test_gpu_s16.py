"""The split-fp16 ("S16") inference path: same parity gates as the exact-fp32 path (1e-4 of
max|ref| on frames and activations, commit scalars 1e-4), against the CPU oracle and the vectors
recorded from the reference."""
import json
import os

import numpy as np
import pytest
import torch

import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import _lib, synthetic as S
from ammcnet_aaai2021_amd.engine import _ptr
from oracle import ammc_oracle as O
from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _net(n_embed=256, k=2):
    sd = S.make_twostream_state(n_embed=n_embed, k=k)
    net = A.get_twostream((12, 6), (3, 2), 64, n_embed, k)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    net.precision = "s16"
    return net, sd


def test_split_roundtrip_is_fp32_accurate():
    """hi + lo*2^-11 reproduces an fp32 tensor to ~2^-22 relative (layout kernels, both directions)"""
    lib = _lib.load()
    x = (S.hashed_normal("rt", (2, 16, 12, 20), 3.0) * torch.logspace(-3, 2, 16).view(1, 16, 1, 1)).to(DEV)
    y = torch.zeros((2, 14, 22, 16), device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    ps, rs = 16, 22 * 16
    assert lib.ammc_nchw_to_s16_f32(_ptr(x), 2, 16, 12, 20, _ptr(y, rs + ps), 14 * rs, rs, ps, 16, s) == 0
    back = torch.empty_like(x)
    assert lib.ammc_s16_to_nchw_f32(_ptr(y, rs + ps), 14 * rs, rs, ps, 2, 16, 12, 20, _ptr(back), s) == 0
    # normal range: 2^-22 relative; below the fp16 normal range (|v| < 6.1e-5) the pair still resolves
    # 2^-12 of the fp16 subnormal spacing, i.e. 1.5e-11 absolute
    assert bool(((back - x).abs() <= 2.0 ** -21 * x.abs() + 2e-11).all())
    assert float(y[:, 0].abs().max()) == 0.0 and float(y[:, :, 0].abs().max()) == 0.0      # halo untouched


def test_split_encoding_bits():
    """the S16 encoder of the kernels (packed convert + mixed-precision fma, ammc_common.h) produces exactly
    hi = fp16(t), lo = fp16((t - hi) * 2048), both round-to-nearest-even: normal range, fp16 subnormals, exact halves
    (ties), the largest encodable values, zeros and negatives"""
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    vals = torch.cat([
        torch.randn(4096, generator=g) * torch.logspace(-9, 4, 4096),
        torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, -65504.0, 65503.99, 6.1e-5, 5.96e-8, 3e-8, 1.00048828125,
                      1.000732421875, 0.333251953125 + 2.0 ** -13, 2049.0, 2050.0, -2051.0, 1e-30]),
        (torch.arange(2048, dtype=torch.float32) + 0.5) * 2.0 ** -11 + 1.0,          # ties of the hi rounding
    ]).float()
    n = (vals.numel() // 8) * 8
    vals = vals[:n].contiguous()
    x = vals.view(1, 8, 1, n // 8).to(DEV)                       # 8 channels, W = n / 8 pixels of one row
    W = n // 8
    y = torch.zeros((1, 3, W + 2, 8), device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    ps, rs = 8, (W + 2) * 8
    assert lib.ammc_nchw_to_s16_f32(_ptr(x), 1, 8, 1, W, _ptr(y, rs + ps), 3 * rs, rs, ps, 8, s) == 0
    got = y[0, 1, 1:W + 1].cpu().contiguous().view(torch.float16).view(W, 2, 8)      # [pixel][hi | lo][channel]
    t = vals.view(8, W).t().contiguous().numpy()                 # [pixel][channel]
    hi = t.astype(np.float16)
    lo = ((t - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    assert np.array_equal(got[:, 0].numpy().view(np.uint16), hi.view(np.uint16))
    assert np.array_equal(got[:, 1].numpy().view(np.uint16), lo.view(np.uint16))


@pytest.mark.parametrize("name", ["twostream_64_b2_eval", "twostream_64_b2_m2000_eval", "twostream_256_b2_eval"])
def test_twostream_s16_vs_oracle_and_golden(name):
    d = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    cfg = json.loads(str(d["cfg"]))
    net, sd = _net(cfg["n_embed"], cfg["k"])
    rgb_x, op_x, rgb_t, _ = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])
    rgb, op, (rd, od), (rq, oq) = net(rgb_x.to(DEV), op_x.to(DEV))
    assert net._engine.precision == "s16"
    rgb, op, rd, od, rq, oq = (t.cpu() for t in (rgb, op, rd, od, rq, oq))
    step = int(d["out_step"])
    errs = dict(rgb=rel_err(rgb[..., ::step, ::step], d["rgb"]), op=rel_err(op[..., ::step, ::step], d["op"]),
                rd=rel_err(rd, d["rgb_diff"]), od=rel_err(od, d["op_diff"]), rq=rel_err(rq, d["rgb_q"]),
                oq=rel_err(oq, d["op_q"]))
    assert max(errs.values()) <= TOL, errs
    if cfg["hw"] <= 64:
        with torch.no_grad():
            w = O.twostream_forward(O.clone_state(sd), rgb_x, op_x, cfg["k"])
        assert rel_err(rgb, w[0]) <= TOL and rel_err(op, w[1]) <= TOL
        eng, st = net._engine, net._engine._last
        r = st["streams"][0]
        for ref_name, act in (("rgb.inc", r.skip[0]), ("rgb.down2", r.skip[2]), ("rgb.down3", r.x4),
                              ("rgb.vq_down3", r.x4q), ("rgb.bridge", st["bridge"][0]), ("op.bridge", st["bridge"][1]),
                              ("rgb.up3", r.u3)):
            want = d[f"st.{ref_name}"]
            got = eng.act_nchw(act).cpu()
            stp = got.shape[-1] // want.shape[-1]
            assert rel_err(got[..., ::stp, ::stp], want) <= TOL, ref_name
        assert net.quant_befor.shape == (2, 512, 8, 8)


def test_s16_and_fp32_paths_agree_and_switch():
    net, sd = _net()
    rgb_x, op_x, _, _ = S.make_clips(3, 64, 64, tag="sw")
    a = net(rgb_x.to(DEV), op_x.to(DEV))
    net.precision = "fp32"
    b = net(rgb_x.to(DEV), op_x.to(DEV))
    assert net._engine.precision == "fp32"
    assert rel_err(a[0].cpu(), b[0].cpu()) <= 2e-5 and rel_err(a[2][0].cpu(), b[2][0].cpu()) <= 2e-5
    net.precision = "s16"
    c = net(rgb_x.to(DEV), op_x.to(DEV))
    assert torch.equal(a[0], c[0])                                   # deterministic


def test_unet_and_unetmem_s16():
    d = np.load(os.path.join(GOLDEN, "unet_64_b2_eval.npz"))
    cfg = json.loads(str(d["cfg"]))
    net = A.get_unet(12, 3)
    net.load_state_dict(S.make_unet_state(12, 3))
    net = net.to(DEV).eval()
    net.precision = "s16"
    x = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])[0]
    assert rel_err(net(x.to(DEV)).cpu(), d["y"]) <= TOL


def _overflowing_state(sd, key="rgb.inc.conv.conv.1.weight", gain=3e5):
    """the synthetic parameters with one BatchNorm gamma blown up: activations of that layer leave the half range"""
    return {k: (v * gain if k == key else v) for k, v in sd.items()}


def test_s16_overflow_guard_recomputes_on_fp32():
    """activations beyond the fp16 range: the S16 model (guard on by DEFAULT) falls back to the exact-fp32 kernels"""
    net, sd = _net()
    big = _overflowing_state(sd)
    net.load_state_dict(big)
    rgb_x, op_x, _, _ = S.make_clips(1, 64, 64, tag="ovf")
    net.s16_guard = False
    net(rgb_x.to(DEV), op_x.to(DEV))
    assert net._engine.overflowed()                                  # unguarded S16 left the half range here
    del net.s16_guard                                                # back to the default: guarded
    out = net(rgb_x.to(DEV), op_x.to(DEV))
    with torch.no_grad():
        w = O.twostream_forward(O.clone_state(big), rgb_x, op_x, 2)
    assert net.s16_fallbacks == 1 and rel_err(out[0].cpu(), w[0]) <= TOL
    # a clean batch afterwards stays on the S16 kernels (the sticky flag was cleared)
    net.load_state_dict(sd)
    net(rgb_x.to(DEV), op_x.to(DEV))
    assert net.s16_fallbacks == 1


def test_s16_overflow_in_a_middle_layer_at_256_through_the_harness():
    """the evaluation loop (harness.evaluate_dataset -> forward_scored, batches queued back to back, ONE copy to the
    host): a BatchNorm gamma that saturates a MIDDLE layer (down2, the 64x64 level, served by the halo-patch kernel at
    256x256) must be caught by the per-batch flags that travel with the scores, and the flagged batches re-run on the
    exact-fp32 kernels - records equal to an all-fp32 evaluation"""
    from ammcnet_aaai2021_amd import harness
    net, sd = _net()
    big = _overflowing_state(sd, "rgb.down2.mpconv.1.conv.4.weight", 1e6)
    net.load_state_dict(big)
    vids = []
    for v in range(2):
        rgb = S.hashed_uniform(f"ovf-vid{v}", (9 + v, 3, 256, 256))
        op = S.hashed_normal(f"ovf-flow{v}", (8 + v, 2, 256, 256), 2.0 / 256.0)
        vids.append((rgb, op))
    got = harness.evaluate_dataset(net, vids, "synthetic", device=DEV)
    assert net.s16_fallbacks == 2 and net._engine.precision == "s16"            # one batch per sub-video, both flagged
    ref, _ = _net()
    ref.load_state_dict(big)
    ref.precision = "fp32"
    want = harness.evaluate_dataset(ref, vids, "synthetic", device=DEV)
    # (equal up to the order of the float atomics that accumulate the squared errors inside the `outc` kernel)
    for key in ("rgb_img_pred_records", "rgb_fea_comm_records", "op_img_pred_records", "op_fea_comm_records"):
        for a, b in zip(got[key], want[key]):
            assert np.allclose(a, b, rtol=2e-6, atol=0), key
    assert all(np.isfinite(r).all() for r in got["rgb_img_pred_records"])
    # and with clean parameters nothing falls back
    net2, _ = _net()
    harness.evaluate_dataset(net2, vids, "synthetic", device=DEV)
    assert getattr(net2, "s16_fallbacks", 0) == 0


def test_fused_maxpool_matches_pool_kernel(monkeypatch):
    """the halo-patch kernel's second output (2x2 max-pool of the layer it just computed) against
    the stand-alone pooling kernel: all three encoder levels fuse at batch 8, 256x256.  The pooled VALUES are the same
    (the S16 rounding is monotone, so it commutes with max); a value on a half-precision rounding tie may get another
    (hi, lo) pair for the same number, which moves later fp32 accumulations in the last bit: 2e-6 of max|ref|."""
    rgb_x, op_x, _, _ = (t.to(DEV) for t in S.make_clips(8, 256, 256, tag="fusepool"))
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("AMMC_FUSE_POOL", flag)
        net, _ = _net()
        with torch.no_grad():
            out = net(rgb_x, op_x)
        names = [m["name"] for m in net._engine._last["plan"].meta]
        assert (sum("pool" in n for n in names) == 0) == (flag == "1")
        outs.append(out)
    a, b = outs
    for x, y in ((a[0], b[0]), (a[1], b[1]), (a[3][0], b[3][0]), (a[3][1], b[3][1])):
        assert float((x - y).abs().max() / y.abs().max()) <= 2e-6
    assert abs(float(a[2][0]) - float(b[2][0])) <= 1e-6 * abs(float(b[2][0]))


def test_forward_is_bitwise_repeatable_at_the_benchmark_size():
    """race detector for the DMA-pipelined kernels (counted vmcnt waits, raw barriers, persistent workgroups): the
    inference forward has no atomics on its frame path, so twelve forwards of the same batch (16 clips, 256x256: every
    kernel instance of the benchmark, eight tiles per persistent workgroup) must give the same bits"""
    net, _ = _net(2000, 2)
    rgb_x, op_x, _, _ = (t.to(DEV) for t in S.make_clips(16, 256, 256, tag="repeat"))
    ref = None
    with torch.no_grad():
        for i in range(12):
            out = net(rgb_x, op_x)
            got = (out[0].clone(), out[1].clone(), out[3][0].clone(), out[3][1].clone())
            if ref is None:
                ref = got
            else:
                for a, b in zip(got, ref):
                    assert torch.equal(a, b), i
    assert getattr(net, "s16_fallbacks", 0) == 0


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (3, 64, 96), (5, 32, 64), (24, 64, 64), (7, 128, 32), (2, 8, 8), (9, 96, 160)])
def test_s16_dispatch_is_consistent_across_shapes(B, H, W):
    """which kernel a layer gets (halo-patch variants, implicit GEMM, split-K, fused pooling) depends on batch and frame
    size; whatever the mix, the S16 model must agree with the exact-fp32 kernels (2e-5 of max|ref|) and the two
    streams' commit scores to 1e-5"""
    net, _ = _net()
    rgb_x, op_x, _, _ = (t.to(DEV) for t in S.make_clips(B, H, W, tag=f"shape-{B}-{H}-{W}"))
    with torch.no_grad():
        a = net(rgb_x, op_x)
        net.precision = "fp32"
        b = net(rgb_x, op_x)
    for x, y in ((a[0], b[0]), (a[1], b[1])):
        assert float((x - y).abs().max() / y.abs().max()) <= 2e-5
    for x, y in zip(a[2], b[2]):
        assert abs(float(x) - float(y)) <= 1e-5 * abs(float(y))


@pytest.mark.parametrize("prec,mf", [("s16", -1), ("s16", 1), ("s16", 0), ("fp32", 1)])
def test_benchmark_workload_b16_256_m2000_vs_reference_vectors(prec, mf):
    """BASELINE.json configs[1] exactly as bench.py runs it (batch 16, 256x256, 2000 slots) against vectors recorded
    from the reference (`twostream.forward`, unet.py:981-1007): frames, commit scalars, quantised maps, per-sample
    PSNR and per-stage activations; the S16 run must be made of the kernel variants the benchmark reports."""
    d = np.load(os.path.join(GOLDEN, "twostream_256_b16_m2000_eval.npz"))
    cfg = json.loads(str(d["cfg"]))
    assert (cfg["batch"], cfg["hw"], cfg["n_embed"]) == (16, 256, 2000)
    net, sd = _net(cfg["n_embed"], cfg["k"])
    net.precision = prec
    rgb_x, op_x, rgb_t, _ = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])
    lib = _lib.load()
    _lib.check(lib.ammc_set_option(b"s16_mf", mf), "set_option")       # MFMA shape of the halo-patch kernel (A/B switch)
    try:
        with torch.no_grad():
            (rgb, op, (rd, od), (rq, oq)), psnr, _ = net.forward_scored(rgb_x.to(DEV), op_x.to(DEV), rgb_t.to(DEV))
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.ammc_set_option(b"s16_mf", -1), "set_option")
    eng, st = net._engine, net._engine._last
    assert eng.precision == prec
    step, qs, rows = int(d["out_step"]), int(d["q_step"]), list(d["st_rows"])
    errs = dict(rgb=rel_err(rgb.cpu()[..., ::step, ::step], d["rgb"]), op=rel_err(op.cpu()[..., ::step, ::step], d["op"]),
                rd=rel_err(rd.cpu(), d["rgb_diff"]), od=rel_err(od.cpu(), d["op_diff"]),
                rq=rel_err(rq.cpu()[:, ::qs, ::qs], d["rgb_q"]), oq=rel_err(oq.cpu()[:, ::qs, ::qs], d["op_q"]),
                psnr=rel_err(psnr.cpu(), d["rgb_psnr"]))
    r, o = st["streams"]
    for ref_name, act in (("rgb.inc", r.skip[0]), ("rgb.down1", r.skip[1]), ("rgb.down2", r.skip[2]), ("rgb.down3", r.x4),
                          ("rgb.vq_down3", r.x4q), ("rgb.bridge", st["bridge"][0]), ("op.bridge", st["bridge"][1]),
                          ("rgb.up3", r.u3), ("op.inc", o.skip[0]), ("op.down3", o.x4), ("op.up3", o.u3)):
        want = d[f"st.{ref_name}"]
        got = eng.act_nchw(act)[rows].cpu()
        stp = got.shape[-1] // want.shape[-1]
        errs[ref_name] = rel_err(got[..., ::stp, ::stp], want)
    assert max(errs.values()) <= TOL, errs
    if prec == "s16":
        kernels = {m["kernel"] for m in st["plan"].meta} | {s.outc_kernel for s in st["streams"]}
        # mf = -1: the default dispatch, i.e. what bench.py runs and reports (the k-half-major pipelines for the 4-wave
        # layers of two rounds and more, the 8-wave 16x16x32 form below); 0 / 1: one MFMA shape forced, tap-by-tap loops
        taps = ({"conv_tap_s16<4, 1, 2, 4, 1, 0, 1>", "conv_tap_s16<4, 1, 2, 2, 1, 0, 1>", "conv_tap_s16<4, 2, 2, 2, 2, 1>"} if mf < 0 else
                {f"conv_tap_s16<4, 1, 2, 4, 1, {mf}>", f"conv_tap_s16<4, 1, 2, 2, 1, {mf}>", f"conv_tap_s16<4, 2, 2, 2, 2, {mf}>"})
        # (round 6: the memory block - enc 1x1, lookup, commit sum, re-encoding, dec 1x1 + residual - is ONE launch)
        assert taps | {"conv_outc_s16", "conv_first_s16", "conv_up_s16<2>", "conv_up_s16<4>", "memory_block_s16"} <= kernels, kernels
        assert not ({"memory_topk_s16", "split_rows"} & kernels), kernels
        assert all(s.first_mid is not None for s in st["streams"])                  # conv_first_s16 took the first layers
        assert not eng.overflowed()


@pytest.mark.parametrize("B,H,W", [(1, 100, 100), (2, 36, 52), (1, 72, 96)])
@pytest.mark.parametrize("prec", ["s16", "fp32"])
def test_frame_sizes_not_divisible_by_8(B, H, W, prec):
    """MaxPool2d(2) floors odd levels and `up.forward` pads the transposed conv's output by one zero row / column at the
    end (reference unet.py:36, 53-56): 100 -> 50 -> 25 -> 12 -> (24 padded to 25) ...  Against the oracle, both
    precisions; the quantised maps have the floored bottleneck size."""
    net, sd = _net()
    net.precision = prec
    rgb_x, op_x, _, _ = S.make_clips(B, H, W, tag=f"odd-{B}-{H}-{W}")
    with torch.no_grad():
        rgb, op, (rd, od), (rq, oq) = net(rgb_x.to(DEV), op_x.to(DEV))
        w = O.twostream_forward(O.clone_state(sd), rgb_x, op_x, 2)
    assert rgb.shape == (B, 3, H, W) and rq.shape == (B, H // 2 // 2 // 2, W // 2 // 2 // 2, 64)
    errs = dict(rgb=rel_err(rgb.cpu(), w[0]), op=rel_err(op.cpu(), w[1]), rd=rel_err(rd.cpu(), w[2][0]),
                od=rel_err(od.cpu(), w[2][1]), rq=rel_err(rq.cpu(), w[3][0]), oq=rel_err(oq.cpu(), w[3][1]))
    assert max(errs.values()) <= TOL, errs


def _memory_s16(embed, x, k):
    """`ammc_memory_topk_fwd_s16` on its own: embed [64, M] fp32, x [..., 64] -> (q_topk, diff, q_one, idx)"""
    from ammcnet_aaai2021_amd.engine import _Packer
    lib = _lib.load()
    d, m = embed.shape
    lead = x.shape[:-1]
    x2 = x.to(DEV).float().contiguous().view(-1, d)
    n = x2.shape[0]
    pk = _Packer(torch.device(DEV), s16=True)
    e = embed.to(DEV).contiguous()
    e_md, enorm = pk.codebook(e)
    e16, flag = pk.codebook_s16(e)
    assert pk.flags.tolist()[flag] == 0                      # (inside the half range: no verdict)
    idx = torch.empty((n, k), device=DEV, dtype=torch.int32)
    qk = torch.empty((n, k * d), device=DEV)
    q1 = torch.empty((n, d), device=DEV)
    nblk = lib.ammc_memory_topk_blocks(n)
    part = torch.empty(nblk, device=DEV)
    diff = torch.empty(1, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ammc_memory_topk_fwd_s16(_ptr(x2), e16.data_ptr(), _ptr(e_md), _ptr(enorm), n, d, m, k, idx.data_ptr(),
                                            _ptr(qk), _ptr(q1), _ptr(part), s), "memory_topk_s16")
    _lib.check(lib.ammc_sum_partials_f32(_ptr(part), nblk, 1.0 / float(n * d), _ptr(diff), s), "sum")
    return qk.view(*lead, k * d).cpu(), diff[0].cpu(), q1.view(*lead, d).cpu(), idx.view(*lead, k).cpu()


@pytest.mark.parametrize("m,k,bhw", [(2000, 2, (4, 16, 17)), (256, 1, (3, 11, 13)), (40, 2, (1, 9, 15))])
def test_memory_topk_s16_row_tiles_agree_bit_for_bit(m, k, bhw):
    """the 64-row form of the S16 memory kernel (8 waves, one slot tile per step; what 16384 rows and more take) against
    the 32-row form on the same ragged inputs: every output identical (same accumulation order per distance, one
    commit partial per 32 rows in both), and against the oracle with the gates of the test below"""
    from test_gpu_parity import _check_quantize
    lib = _lib.load()
    embed = S.hashed_normal(f"gq2:64:{m}", (64, m), 0.9)
    x = S.hashed_normal(f"gq2x:64:{m}", (*bhw, 64), 0.8)
    outs = {}
    try:
        for rt in (1, 2):
            assert lib.ammc_set_option(b"memory_rt", rt) == 0
            outs[rt] = _memory_s16(embed, x, k)
    finally:
        lib.ammc_set_option(b"memory_rt", 0)
    for a, b in zip(outs[1], outs[2]):
        assert torch.equal(a, b)
    _check_quantize(x, embed, k, *outs[2])
    assert lib.ammc_set_option(b"memory_rt", 3) == -1


@pytest.mark.parametrize("m,k,bhw", [(256, 2, (2, 8, 8)), (2000, 2, (3, 8, 8)), (256, 3, (1, 4, 8)), (256, 1, (1, 5, 7)),
                                     (5000, 2, (1, 3, 11)), (33, 4, (2, 2, 2))])
def test_memory_topk_s16(m, k, bhw):
    """the S16 memory kernel (distance GEMM on the fp16 MFMA pipe, fp32-equivalent; the inference default at
    embed_dim 64) against the oracle with the SAME gates as the exact-fp32 kernel (tests/test_gpu_parity.py): indices
    equal wherever the margin exceeds fp32 noise, gathered rows bit exact, commit distance 1e-4; ragged n and m, more
    slots than the norm cache holds"""
    from test_gpu_parity import _check_quantize
    embed = S.hashed_normal(f"gq:64:{m}", (64, m), 0.9)
    x = S.hashed_normal(f"gqx:64:{m}", (*bhw, 64), 0.8)
    qk, diff, q1, idx = _memory_s16(embed, x, k)
    _check_quantize(x, embed, k, qk, diff, q1, idx)
    assert _lib.load().ammc_memory_topk_fwd_s16(1, 1, 1, 1, 8, 128, m, k, 1, 1, 1, 1, None) == -2      # d != 64: the fp32 entry


def test_memory_topk_s16_golden_and_tie():
    g = np.load(os.path.join(GOLDEN, "quantize_cases.npz"))
    for cname in ("m256", "m2000", "k3"):
        c = json.loads(str(g[f"{cname}.cfg"]))
        embed = S.hashed_normal(f"quantize_cases:{cname}:embed", (c["d"], c["m"]), 0.9)
        x = S.hashed_normal(f"quantize_cases:{cname}:x", (*c["bhw"], c["d"]), 0.8)
        qk, diff, _, _ = _memory_s16(embed, x, c["k"])
        same = (qk.numpy() == g[f"{cname}.qk"]).all(axis=-1)
        assert same.mean() > 0.98, cname
        assert rel_err(diff, g[f"{cname}.diff"]) <= TOL
    embed = S.hashed_normal("quantize_cases:tie:embed", (64, 256), 0.9)
    qk, _, _, _ = _memory_s16(embed, torch.from_numpy(g["tie.x"]), 2)
    assert np.array_equal(qk.numpy(), g["tie.qk"])                   # 1e-3 off the bisector: unambiguous


@pytest.mark.parametrize("B,H,W,m", [(2, 64, 64, 256), (3, 72, 96, 2000), (16, 256, 256, 2000), (1, 256, 256, 33)])
def test_memory_block_as_one_launch_equals_the_five_launch_chain(B, H, W, m):
    """`ammc_memory_block_s16` (round 6: enc 1x1 -> distances + top-2 -> gather -> dec 1x1 + residual in one kernel, the
    commit sum by the last workgroup) against the chain it replaces (conv_gemm_s16 1x1, memory_topk_s16, sum_partials,
    split_rows, conv_gemm_s16 1x1): the same k order and expressions, so EVERYTHING is bit-identical - frames, commit
    scalars, quantised maps, lookups, the bottleneck after the block.  324 rows (72x96): a ragged last 64-row tile; 33
    slots: a single, ragged slot tile; twice in a row: the arrival counter is left at zero."""
    from ammcnet_aaai2021_amd import engine as E
    sd = S.make_twostream_state(n_embed=m)
    rgb_x, op_x, _, _ = (t.to(DEV) for t in S.make_clips(B, H, W, tag=f"fused-memory:{B}:{H}"))
    outs = {}
    old = E.FUSED_MEMORY
    try:
        for fused in (True, False):
            E.FUSED_MEMORY = fused
            net = A.get_twostream((12, 6), (3, 2), 64, m, 2)
            net.load_state_dict(sd)
            net = net.to(DEV).eval()
            net.precision = "s16"
            with torch.no_grad():
                for _ in range(2):
                    rgb, op, (rd, od), (rq, oq) = net(rgb_x, op_x)
            eng = net._last_engine
            st = eng._last
            kernels = {mm["kernel"] for mm in st["plan"].meta}
            assert ("memory_block_s16" in kernels) == fused and ("memory_topk_s16" in kernels) == (not fused), kernels
            outs[fused] = dict(rgb=rgb.clone(), op=op.clone(), rd=rd.clone(), od=od.clone(), rq=rq.clone(), oq=oq.clone(),
                               idx=[s_.idx.clone() for s_ in st["streams"]],
                               x4q=[eng.act_nchw(s_.x4q).clone() for s_ in st["streams"]])
            assert not eng.overflowed()
    finally:
        E.FUSED_MEMORY = old
    a, b = outs[True], outs[False]
    for key in ("rgb", "op", "rd", "od", "rq", "oq"):
        assert torch.equal(a[key], b[key]), key
    for i in range(2):
        assert torch.equal(a["idx"][i], b["idx"][i]) and torch.equal(a["x4q"][i], b["x4q"][i]), i
