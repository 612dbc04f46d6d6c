"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden vectors
recorded from the reference.  Needs an MI355X: run with `-m gpu`.

Whole-model tests here pin `precision = "fp32"` (the exact-fp32 MFMA kernels); the package default, S16, has the same
gates in tests/test_gpu_s16.py.

Tolerances (BASELINE.json north_star):
  predicted frames / activations   max|d| / max|ref| <= 1e-4   (fp32 MFMA, exact-fp32 products)
  commit scalars                   rel <= 1e-4
  gathered codebook rows           bit exact wherever the slot choice is unambiguous
"""
import json
import os

import numpy as np
import pytest
import torch

import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import _lib, ops, synthetic as S
from oracle import ammc_oracle as O
from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _gold(name):
    d = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    return d, json.loads(str(d["cfg"]))


def _twostream(n_embed=256, k=2, tag="ammc"):
    sd = S.make_twostream_state(n_embed=n_embed, k=k, tag=tag)
    net = A.get_twostream((12, 6), (3, 2), 64, n_embed, k)
    net.load_state_dict(sd, strict=True)
    net.precision = "fp32"                 # this file pins the exact-fp32 kernels; tests/test_gpu_s16.py the default (S16)
    return net.to(DEV).eval(), sd


def test_library_is_the_hip_build():
    assert b"gfx950" in _lib.load().ammc_build_info()
    assert torch.cuda.is_available()


# ---- per-kernel parity -----------------------------------------------------------------

@pytest.mark.parametrize("cin,cout,hw,batch", [(12, 64, 32, 2), (6, 64, 24, 1), (64, 128, 16, 3),
                                               (128, 256, 8, 2), (512, 512, 8, 2), (256, 64, 40, 1)])
def test_double_conv(cin, cout, hw, batch):
    """K1: conv3x3 + folded BN + ReLU, both tile shapes, ragged M (partial last tile)"""
    dc = A.double_conv(cin, cout)
    sd = {k: S._fill(f"x.{k}", tuple(v.shape), "dc") for k, v in dc.state_dict().items()}
    dc.load_state_dict(sd)
    dc = dc.to(DEV).eval()
    x = S.hashed_uniform(f"dcx{cin}", (batch, cin, hw, hw + 8))
    want = O.double_conv({f"p.{k[5:]}": v for k, v in sd.items()}, "p", x)
    got = dc(x.to(DEV)).cpu()
    assert rel_err(got, want) <= TOL


def test_down_and_maxpool():
    m = A.down(64, 128)
    sd = {k: S._fill(f"x.{k}", tuple(v.shape), "dn") for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = S.hashed_uniform("downx", (2, 64, 32, 48))
    want = O.down({f"p.{k}": v for k, v in sd.items()}, "p", x)
    assert rel_err(m(x.to(DEV)).cpu(), want) <= TOL


@pytest.mark.parametrize("c", [128, 256, 512])
def test_up_convtranspose_concat(c):
    """K3/K4: ConvTranspose 2x2 s2 scattered into the concat buffer + double_conv"""
    m = A.up(c, c // 2)
    sd = {k: S._fill(f"x.{k}", tuple(v.shape), "up") for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x1 = S.hashed_uniform("upx1", (2, c, 8, 12))
    x2 = S.hashed_uniform("upx2", (2, c // 2, 16, 24))
    want = O.up({f"p.{k}": v for k, v in sd.items()}, "p", x1, x2)
    assert rel_err(m(x1.to(DEV), x2.to(DEV)).cpu(), want) <= TOL


def test_bridge_amft():
    m = A.bridge(512)
    sd = {k: S._fill(f"bridge.{k}", tuple(v.shape), "br") for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    zx = S.hashed_uniform("zx", (2, 512, 8, 8))
    zy = S.hashed_uniform("zy", (2, 512, 8, 8))
    wx, wy = O.bridge({f"bridge.{k}": v for k, v in sd.items()}, zx, zy)
    gx, gy = m(zx.to(DEV), zy.to(DEV))
    assert rel_err(gx.cpu(), wx) <= TOL and rel_err(gy.cpu(), wy) <= TOL


def _check_quantize(x, embed, k, qk, diff, q1, idx):
    """compare against the oracle; rows whose k-th / (k+1)-th distances are closer than the
    fp32 noise of the expanded distance formula may legitimately pick either slot"""
    wqk, wdiff, widx, widx1, flat, wq1 = O.quantize_topk(x, embed, k)
    d = embed.shape[0]
    dist = (flat.double().pow(2).sum(1, keepdim=True) - 2 * flat.double() @ embed.double()
            + embed.double().pow(2).sum(0, keepdim=True))
    srt = dist.sort(dim=1).values
    margin = (srt[:, 1:k + 1] - srt[:, :k]).min(dim=1).values
    scale = flat.double().pow(2).sum(1) + 1.0
    safe = margin > 1e-4 * scale
    assert safe.float().mean() > 0.9
    idx = idx.reshape(-1, k).long()
    assert torch.equal(idx[safe], widx.reshape(-1, k)[safe])
    assert torch.equal(qk.reshape(-1, k * d)[safe], wqk.reshape(-1, k * d)[safe])       # gather: bit exact
    assert rel_err(q1.reshape(-1, d)[safe], (x + (wq1 - x)).reshape(-1, d)[safe]) <= 1e-6
    assert rel_err(diff, wdiff) <= TOL
    # every chosen slot is a true nearest neighbour up to fp32 noise, even on unsafe rows
    chosen = dist.gather(1, idx)
    assert bool(((chosen - srt[:, :k]).abs() <= 1e-4 * scale[:, None]).all())


@pytest.mark.parametrize("d,m,k,bhw", [(64, 256, 2, (2, 8, 8)), (64, 2000, 2, (3, 8, 8)), (64, 256, 3, (1, 4, 8)),
                                       (64, 256, 1, (1, 5, 7)), (128, 1000, 4, (2, 9, 7)), (256, 512, 2, (1, 8, 8))])
def test_memory_topk(d, m, k, bhw):
    """K6: fused distance GEMM + running top-k + gather + commit distance; ragged n and m"""
    q = A.Quantize_topk(d, m, k=k)
    embed = S.hashed_normal(f"gq:{d}:{m}", (d, m), 0.9)
    q.embed.copy_(embed)
    q = q.to(DEV).eval()
    x = S.hashed_normal(f"gqx:{d}:{m}", (*bhw, d), 0.8)
    qk, diff, q1 = q(x.to(DEV))
    _check_quantize(x, embed, k, qk.cpu(), diff.cpu(), q1.cpu(), q.last_indices.cpu())


def test_memory_topk_golden_and_tie():
    g = np.load(os.path.join(GOLDEN, "quantize_cases.npz"))
    for cname in ("m256", "m2000", "k3"):
        c = json.loads(str(g[f"{cname}.cfg"]))
        q = A.Quantize_topk(c["d"], c["m"], k=c["k"])
        q.embed.copy_(S.hashed_normal(f"quantize_cases:{cname}:embed", (c["d"], c["m"]), 0.9))
        q = q.to(DEV).eval()
        x = S.hashed_normal(f"quantize_cases:{cname}:x", (*c["bhw"], c["d"]), 0.8)
        qk, diff, q1 = q(x.to(DEV))
        same = (qk.cpu().numpy() == g[f"{cname}.qk"]).all(axis=-1)
        assert same.mean() > 0.98, cname                   # near-ties may differ; see _check_quantize
        assert rel_err(diff.cpu(), g[f"{cname}.diff"]) <= TOL
    # the deliberate near-tie fixture: x is 1e-3 off the bisector of two slots -> unambiguous in fp32
    q = A.Quantize_topk(64, 256, k=2)
    q.embed.copy_(S.hashed_normal("quantize_cases:tie:embed", (64, 256), 0.9))
    q = q.to(DEV).eval()
    qk, _, _ = q(torch.from_numpy(g["tie.x"]).to(DEV))
    assert np.array_equal(qk.cpu().numpy(), g["tie.qk"])


def test_vq_block_residual():
    m = A.enc_quan_dec_res_topk(512, 64, 256, k=2)
    sd = {k: S._fill(f"vq_down3.{k}", tuple(v.shape), "vq") for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = S.hashed_uniform("vqx", (2, 512, 8, 8), 0.0, 2.0)
    want, wdiff, wq, _ = O.vq_block({f"vq_down3.{k}": v for k, v in sd.items()}, "vq_down3", x, 2)
    got, diff, q1 = m(x.to(DEV))
    assert rel_err(got.cpu(), want) <= TOL and rel_err(diff.cpu(), wdiff) <= TOL and diff.shape == (1,)
    assert rel_err(q1.cpu(), wq) <= TOL


# ---- whole models ------------------------------------------------------------------------

def _check_twostream(net, sd, cfg, d=None, full_oracle=True):
    rgb_x, op_x, rgb_t, _ = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])
    with torch.no_grad():
        rgb, op, (rd, od), (rq, oq) = net(rgb_x.to(DEV), op_x.to(DEV))
    rgb, op, rd, od, rq, oq = (t.cpu() for t in (rgb, op, rd, od, rq, oq))
    assert rgb.shape == (cfg["batch"], 3, cfg["hw"], cfg["hw"]) and rd.shape == (1,)
    if full_oracle:
        with torch.no_grad():
            w = O.twostream_forward(O.clone_state(sd), rgb_x, op_x, cfg["k"])
        assert rel_err(rgb, w[0]) <= TOL and rel_err(op, w[1]) <= TOL
        assert rel_err(rd, w[2][0]) <= TOL and rel_err(od, w[2][1]) <= TOL
        assert rel_err(rq, w[3][0]) <= TOL and rel_err(oq, w[3][1]) <= TOL
    if d is not None:
        step = int(d["out_step"])
        assert rel_err(rgb[..., ::step, ::step], d["rgb"]) <= TOL
        assert rel_err(op[..., ::step, ::step], d["op"]) <= TOL
        assert rel_err(rd, d["rgb_diff"]) <= TOL and rel_err(od, d["op_diff"]) <= TOL
        assert rel_err(rq, d["rgb_q"]) <= TOL and rel_err(oq, d["op_q"]) <= TOL
        psnr = torch.stack([O.psnr_error(rgb[i:i + 1], rgb_t[i:i + 1]) for i in range(rgb.shape[0])])
        assert rel_err(psnr, d["rgb_psnr"]) <= 1e-4


@pytest.mark.parametrize("name", ["twostream_64_b2_eval", "twostream_64_b2_m2000_eval"])
def test_twostream_small_vs_oracle_and_golden(name):
    d, cfg = _gold(name)
    net, sd = _twostream(cfg["n_embed"], cfg["k"])
    _check_twostream(net, sd, cfg, d)
    # the reference's side-effect attributes (unet.py:986,988)
    assert net.quant_befor.shape == (2, 512, 8, 8) and net.quant_after.shape == (2, 512, 8, 8)
    # stage-level parity against the recorded reference activations
    st = net._engine._last
    r = st["streams"][0]
    for ref_name, act in (("rgb.inc", r.skip[0]), ("rgb.down1", r.skip[1]), ("rgb.down2", r.skip[2]),
                          ("rgb.down3", r.x4), ("rgb.vq_down3", r.x4q), ("rgb.bridge", st["bridge"][0]),
                          ("op.bridge", st["bridge"][1]), ("rgb.up3", r.u3)):
        want = d[f"st.{ref_name}"]
        got = act.interior().permute(0, 3, 1, 2).cpu()
        stp = got.shape[-1] // want.shape[-1]
        assert rel_err(got[..., ::stp, ::stp], want) <= TOL, ref_name


def test_twostream_256_golden():
    """BASELINE frame size; compared with the vectors recorded from the reference"""
    d, cfg = _gold("twostream_256_b2_eval")
    net, sd = _twostream(cfg["n_embed"], cfg["k"])
    _check_twostream(net, sd, cfg, d, full_oracle=False)


def test_twostream_repeat_is_bit_identical_and_batch_consistent():
    """size-independent properties: determinism, and clips do not interact except in the commit mean"""
    net, sd = _twostream(256, 2)
    rgb_x, op_x, _, _ = S.make_clips(3, 64, 64, tag="prop")
    a = net(rgb_x.to(DEV), op_x.to(DEV))
    b = net(rgb_x.to(DEV), op_x.to(DEV))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2][0], b[2][0])
    one = net(rgb_x[1:2].to(DEV), op_x[1:2].to(DEV))
    assert rel_err(one[0].cpu(), a[0][1:2].cpu()) <= 1e-6
    singles = [net(rgb_x[i:i + 1].to(DEV), op_x[i:i + 1].to(DEV))[2][0] for i in range(3)]
    assert rel_err(torch.stack(singles).mean().cpu(), a[2][0].cpu()[0]) <= 1e-5


def test_weights_are_repacked_after_load_state_dict():
    net, sd = _twostream(256, 2)
    rgb_x, op_x, _, _ = S.make_clips(1, 64, 64, tag="repack")
    a = net(rgb_x.to(DEV), op_x.to(DEV))[0].clone()
    sd2 = S.make_twostream_state(tag="other")
    net.load_state_dict(sd2)
    b = net(rgb_x.to(DEV), op_x.to(DEV))[0]
    with torch.no_grad():
        w = O.twostream_forward(O.clone_state(sd2), rgb_x, op_x, 2)
    assert not torch.equal(a, b) and rel_err(b.cpu(), w[0]) <= TOL


def test_unet_config1_and_unetmem():
    d, cfg = _gold("unet_64_b2_eval")
    net = A.get_unet(12, 3)
    net.load_state_dict(S.make_unet_state(12, 3))
    net = net.to(DEV).eval()
    net.precision = "fp32"
    x = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])[0]
    assert rel_err(net(x.to(DEV)).cpu(), d["y"]) <= TOL
    sd = {k[4:]: v for k, v in S.make_twostream_state().items() if k.startswith("rgb.")}
    one = A.get_unet_vq_topk_res(12, 3, 64, 256, 2)
    one.load_state_dict(sd)
    one = one.to(DEV).eval()
    one.precision = "fp32"
    y, diff, q1 = one(x.to(DEV))
    with torch.no_grad():
        wy, wd, wq = O.unetmem_forward(O.clone_state(sd), x, 2)
    assert rel_err(y.cpu(), wy) <= TOL and rel_err(diff.cpu(), wd) <= TOL and rel_err(q1.cpu(), wq) <= TOL


def test_eval_records_through_hip_path():
    """the scoring loop either side of the path (test_helper.py:408-473) driven by the HIP model"""
    net, sd = _twostream(256, 2)
    t = 23
    rgb = S.hashed_uniform("vid", (t, 3, 64, 64))
    u = S.hashed_normal("vidop", (t - 1, 1, 64, 64), 2.0) / 256.0
    op = torch.cat([u, u / 256.0], 1)
    rec_hip = O.eval_subvideo_records(lambda a, b: tuple(_cpu(o) for o in net(a.to(DEV), b.to(DEV))), rgb, op)
    rec_cpu = O.eval_subvideo_records(lambda a, b: O.twostream_forward(O.clone_state(sd), a, b, 2), rgb, op)
    for key in ("rgb_psnr", "rgb_comm", "op_comm"):
        assert rel_err(rec_hip[key], rec_cpu[key]) <= TOL, key


def _cpu(o):
    if isinstance(o, tuple):
        return tuple(_cpu(x) for x in o)
    return o.cpu()


def test_harness_dataset_eval_on_hip_model():
    """the build's scoring loop + score fusion, HIP model vs CPU oracle, two sub-videos"""
    from ammcnet_aaai2021_amd import harness as Hn
    net, sd = _twostream(256, 2)
    vids = []
    for i, t in enumerate((21, 38)):
        rgb = S.hashed_uniform(f"hv{i}", (t, 3, 64, 64))
        u = S.hashed_normal(f"ho{i}", (t - 1, 1, 64, 64), 2.0) / 256.0
        vids.append((rgb, torch.cat([u, u / 256.0], 1)))
    rec = Hn.evaluate_dataset(net, vids, "ped2", device=DEV)
    want = Hn.evaluate_dataset(lambda a, b: O.twostream_forward(O.clone_state(sd), a, b, 2), vids, "ped2")
    for key in ("rgb_img_pred_records", "rgb_fea_comm_records", "op_fea_comm_records"):
        for a, b in zip(rec[key], want[key]):
            assert rel_err(a, b) <= TOL, key
    gt = [(S.hashed_uniform(f"gt{i}", (r.shape[0],), 0, 1) > 0.7).numpy().astype(np.int8) for i, (r, _) in enumerate(vids)]
    assert abs(Hn.fuse_scores_auc(rec, gt)["auc_raw"] - Hn.fuse_scores_auc(want, gt)["auc_raw"]) <= 1e-3


@pytest.mark.parametrize("prec", ["fp32", "s16"])
def test_fused_psnr_in_outc_epilogue(prec):
    """per-sample PSNR from the squared error accumulated inside the `outc` kernel"""
    net, sd = _twostream(256, 2)
    net.precision = prec
    rgb_x, op_x, rgb_t, op_t = S.make_clips(3, 64, 72, tag="psnr")       # 64*72 = 36 tiles of 128 pixels per sample
    out, rp, opp = net.forward_scored(rgb_x.to(DEV), op_x.to(DEV), rgb_t.to(DEV), op_t.to(DEV))
    with torch.no_grad():
        w = O.twostream_forward(O.clone_state(sd), rgb_x, op_x, 2)
    want_r = torch.stack([O.psnr_error(w[0][i:i + 1], rgb_t[i:i + 1]) for i in range(3)])
    want_o = torch.stack([O.psnr_error(w[1][i:i + 1], op_t[i:i + 1]) for i in range(3)])
    assert rel_err(rp.cpu(), want_r) <= 1e-5 and rel_err(opp.cpu(), want_o) <= 1e-5
    assert rel_err(out[0].cpu(), w[0]) <= TOL
    # a frame size whose samples are not a whole number of tiles: tiles straddle two samples
    rgb_x, op_x, rgb_t, op_t = S.make_clips(3, 40, 40, tag="psnr2")
    out, rp, _ = net.forward_scored(rgb_x.to(DEV), op_x.to(DEV), rgb_t.to(DEV))
    with torch.no_grad():
        w = O.twostream_forward(O.clone_state(sd), rgb_x, op_x, 2)
    want_r = torch.stack([O.psnr_error(w[0][i:i + 1], rgb_t[i:i + 1]) for i in range(3)])
    assert rel_err(rp.cpu(), want_r) <= 1e-5


def test_hipgraph_replay_matches_eager():
    """`engine.use_graph`: one captured hipGraph per (shape, target pattern); replays must reproduce the eager
    launches bit for bit, with and without the fused PSNR targets, across changing inputs"""
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(S.make_twostream_state())
    net = net.to(DEV).eval()
    for tag in ("graph-a", "graph-b"):
        rgb_x, op_x, rgb_t, op_t = (t.to(DEV) for t in S.make_clips(2, 64, 64, tag=tag))
        want = net(rgb_x, op_x)
        eng = net._engine
        want_s = net.forward_scored(rgb_x, op_x, rgb_t, op_t)
        eng.use_graph = True
        got = net(rgb_x, op_x)
        got_s = net.forward_scored(rgb_x, op_x, rgb_t, op_t)
        eng.use_graph = False
        assert net._engine is eng
        for a, b in ((want[0], got[0]), (want[1], got[1]), (want[2][0], got[2][0]), (want[2][1], got[2][1]),
                     (want[3][0], got[3][0]), (want[3][1], got[3][1]), (want_s[0][0], got_s[0][0])):
            assert torch.equal(a, b)
        # the squared-error accumulation uses float atomics across workgroups: order-dependent in the last bits
        assert torch.allclose(want_s[1], got_s[1], rtol=1e-5, atol=0)
        assert torch.allclose(want_s[2], got_s[2], rtol=1e-5, atol=0)
