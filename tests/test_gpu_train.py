"""Training-mode parity of the HIP path: one optimisation step's forward outputs, every
parameter gradient and the in-place buffer updates (BatchNorm running statistics, EMA
codebook), against the CPU oracle with autograd and the vectors recorded from the reference.

Tolerances: outputs 1e-4 of max|ref|; buffers 1e-4; gradients per tensor, L2-relative:
  max over tensors <= 1e-2, median <= 2e-3, and the tensors downstream of no ReLU-kink flip
  agree to ~1e-5.  SURVEY.md 8(d) guessed 1e-3; the fixture itself is noisier than that: the
  reference's OWN fp32 and fp64 gradients differ by 2e-3 on the rgb decoder / bridge.O2F
  tensors (a pre-activation within fp32 noise of 0 flips its ReLU mask, which moves a whole
  dbeta entry), measured with the oracle in both precisions."""
import json
import os

import numpy as np
import pytest
import torch

import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
from oracle import ammc_oracle as O
from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GTOL = 1e-2          # per-tensor ceiling (see the module docstring)
GMED = 2e-3          # median over tensors


def _l2rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _train_step(net, sd, batch, hw, tag, k=2):
    rgb_x, op_x, rgb_t, op_t = S.make_clips(batch, hw, hw, tag=tag)
    net.train()
    out = net(rgb_x.to(DEV), op_x.to(DEV))
    loss = O.generator_loss(out, rgb_t.to(DEV), op_t.to(DEV))
    loss.backward()
    msd = O.clone_state(sd, requires_grad=True)
    want = O.twostream_forward(msd, rgb_x, op_x, k, training=True)
    wloss = O.generator_loss(want, rgb_t, op_t)
    wloss.backward()
    return out, loss, want, wloss, msd


def test_twostream_train_step_vs_oracle_and_golden():
    d = np.load(os.path.join(GOLDEN, "twostream_64_b2_train.npz"))
    cfg = json.loads(str(d["cfg"]))
    sd = S.make_twostream_state()
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.to(DEV)
    out, loss, want, wloss, msd = _train_step(net, sd, cfg["batch"], cfg["hw"], cfg["tag"])
    assert rel_err(out[0].detach().cpu(), want[0]) <= 1e-4 and rel_err(out[1].detach().cpu(), want[1]) <= 1e-4
    assert rel_err(out[0].detach().cpu(), d["rgb"]) <= 1e-4 and rel_err(out[1].detach().cpu(), d["op"]) <= 1e-4
    assert rel_err(out[2][0].detach().cpu(), want[2][0]) <= 1e-4 and rel_err(out[2][1].detach().cpu(), want[2][1]) <= 1e-4
    assert abs(float(loss) - float(d["loss"])) <= 1e-4 * abs(float(d["loss"]))
    bad, errs = [], []
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        e = _l2rel(p.grad.cpu(), msd[name].grad)
        errs.append(e)
        if e > GTOL:
            bad.append((name, e))
        # golden: norms recorded from the reference's own autograd
        gn = float(d[f"gn.{name}"])
        assert abs(float(p.grad.double().norm()) - gn) <= 2e-3 * gn + 1e-10, name
    assert not bad, bad
    assert float(np.median(errs)) <= GMED and min(errs) <= 1e-5, (np.median(errs), min(errs))
    # buffers updated inside forward: BN running stats, num_batches_tracked, EMA codebook
    nsd = net.state_dict()
    for key, v in msd.items():
        if key in dict(net.named_parameters()):
            continue
        assert rel_err(nsd[key].cpu().double(), v.double()) <= 1e-4, key
    for key in d.files:
        if key.startswith("buf."):
            assert rel_err(nsd[key[4:]].cpu().double(), d[key].astype(np.float64)) <= 1e-4, key


def test_eval_after_train_uses_updated_buffers():
    """the eval plans must be re-packed after a training forward changed BN stats / codebook"""
    sd = S.make_twostream_state()
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.to(DEV)
    rgb_x, op_x, _, _ = S.make_clips(2, 64, 64, tag="tr-ev")
    net.eval()
    before = net(rgb_x.to(DEV), op_x.to(DEV))[0].clone()
    net.train()
    with torch.no_grad():
        net(rgb_x.to(DEV), op_x.to(DEV))
    net.eval()
    after = net(rgb_x.to(DEV), op_x.to(DEV))[0]
    msd = O.clone_state(sd)
    with torch.no_grad():
        O.twostream_forward(msd, rgb_x, op_x, 2, training=True)
        want = O.twostream_forward(msd, rgb_x, op_x, 2)
    assert not torch.equal(before, after)
    assert rel_err(after.cpu(), want[0]) <= 1e-4


def test_unetmem_and_unet_train_step():
    full = S.make_twostream_state()
    sd = {k[4:]: v for k, v in full.items() if k.startswith("rgb.")}
    net = A.get_unet_vq_topk_res(12, 3, 64, 256, 2)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x, _, t, _ = S.make_clips(2, 32, 48, tag="um")
    y, diff, q1 = net(x.to(DEV))
    loss = O.intensity_l2(y, t.to(DEV)) + diff.sum()
    loss.backward()
    msd = O.clone_state(sd, requires_grad=True)
    wy, wd, wq = O.unetmem_forward(msd, x, 2, training=True)
    (O.intensity_l2(wy, t) + wd.sum()).backward()
    assert rel_err(y.detach().cpu(), wy) <= 1e-4 and rel_err(diff.detach().cpu(), wd) <= 1e-4
    bad = [(n, _l2rel(p.grad.cpu(), msd[n].grad)) for n, p in net.named_parameters()
           if _l2rel(p.grad.cpu(), msd[n].grad) > GTOL]
    assert not bad, bad
    # plain UNet (config 1 model) in training mode
    usd = S.make_unet_state(12, 3)
    u = A.get_unet(12, 3)
    u.load_state_dict(usd)
    u = u.to(DEV).train()
    yy = u(x.to(DEV))
    O.intensity_l2(yy, t.to(DEV)).backward()
    m2 = O.clone_state(usd, requires_grad=True)
    wy2 = O.unet_forward(m2, x, training=True)
    O.intensity_l2(wy2, t).backward()
    assert rel_err(yy.detach().cpu(), wy2) <= 1e-4
    bad = [(n, _l2rel(p.grad.cpu(), m2[n].grad)) for n, p in u.named_parameters()
           if _l2rel(p.grad.cpu(), m2[n].grad) > GTOL]
    assert not bad, bad


def test_adam_steps_track_the_oracle():
    """three optimiser steps end to end (weights change -> filters are re-packed every step)"""
    sd = S.make_twostream_state()
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    msd = O.clone_state(sd, requires_grad=True)
    params = [v for v in msd.values() if v.requires_grad]
    wopt = torch.optim.Adam(params, lr=1e-4)
    losses, wlosses = [], []
    for step in range(3):
        rgb_x, op_x, rgb_t, op_t = S.make_clips(2, 64, 64, tag=f"adam{step}")
        opt.zero_grad()
        loss = O.generator_loss(net(rgb_x.to(DEV), op_x.to(DEV)), rgb_t.to(DEV), op_t.to(DEV))
        loss.backward()
        opt.step()
        wopt.zero_grad()
        wl = O.generator_loss(O.twostream_forward(msd, rgb_x, op_x, 2, training=True), rgb_t, op_t)
        wl.backward()
        wopt.step()
        losses.append(float(loss))
        wlosses.append(float(wl))
    assert np.allclose(losses, wlosses, rtol=2e-4), (losses, wlosses)


def test_train_step_at_sizes_the_halo_patch_kernels_take():
    """batch 4 at 128x128: the 128x128 and 64x64 levels go through conv_tap_s16 (fp32 outputs, fp32 residual of the
    input-gradient convs) and wgrad_tap_s16, which the 64x64 fixtures above are too small to reach; compared with
    the oracle's autograd like the first test (same gates)."""
    sd = S.make_twostream_state()
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.to(DEV)
    out, loss, want, wloss, msd = _train_step(net, sd, 4, 128, "train-128")
    assert rel_err(out[0].detach().cpu(), want[0]) <= 1e-4 and rel_err(out[1].detach().cpu(), want[1]) <= 1e-4
    assert abs(float(loss) - float(wloss)) <= 1e-4 * abs(float(wloss))
    errs = []
    for name, p in net.named_parameters():
        ref = msd[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            continue
        errs.append(_l2rel(p.grad.cpu(), ref))
    errs = np.array(errs)
    # (ReLU-kink noise as above; on this fixture the exact-fp32 kernels land at 7e-3 / 2.7e-3, the S16 ones at 5e-3 / 1.5e-3)
    assert errs.max() <= GTOL and np.median(errs) <= 2 * GMED and errs.min() <= 1e-5, (errs.max(), np.median(errs), errs.min())
    nsd = net.state_dict()
    for key, v in msd.items():
        if key not in dict(net.named_parameters()):
            assert rel_err(nsd[key].cpu().double(), v.double()) <= 1e-4, key
