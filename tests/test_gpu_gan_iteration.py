"""The joint G / D iteration of the reference's training loop (run_helper/train_helper.py:296-339) end to end on the HIP
path - generator forward, two FlowNet2-SD forwards, three discriminator forwards, the D backward, the G backward through
D - against vectors recorded from the reference's own generator, discriminator, FlowNet2-SD and loss classes at the
training benchmark's frame size (tests/golden/gan_256_b{2,32}_iteration.npz, `make_golden.py gan <batch>`).  Losses
against the recorded values (1e-4); the gradients of both networks against the fp64 TRUTH of the iteration on the branch
the HIP evaluation took (tests/truth.py: the oracle's generator + `pixel_discriminator` + `flownet2sd_forward` +
`generator_loss_full` in float64 on the device, memory lookups forced to the evaluation's), gated by what the reference's
own recorded fp32 gradients are away from the truth on ITS branch.  bench.py's `train_gan` leg makes the same comparison
(`train_gan.parity.vs_fp64`) from its timed models' first iteration."""
import json
import os

import numpy as np
import pytest
import torch

import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import harness as Hn, synthetic as S
from conftest import GOLDEN
import truth as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _iteration(batch, flow_precision):
    d = np.load(os.path.join(GOLDEN, f"gan_256_b{batch}_iteration.npz"))
    cfg = json.loads(str(d["cfg"]))
    B = cfg["batch"]
    sd_g, sd_d, sd_f = S.make_twostream_state(), S.make_discriminator_state(), S.make_flownet2sd_state()
    G = A.get_twostream((12, 6), (3, 2), 64, cfg["n_embed"], cfg["k"])
    G.load_state_dict(sd_g)
    G = G.to(DEV).train()
    D = A.PixelDiscriminator(3, [128, 256, 512, 512])
    D.load_state_dict(sd_d)
    D = D.to(DEV).train()
    F2 = A.FlowNet2SD()
    F2.load_state_dict(sd_f)
    F2 = F2.to(DEV).eval()
    F2.precision = flow_precision
    # learning rate 0: the iteration runs as written (both optimizer steps included) and leaves parameters - and the
    # gradients of its two backward passes - in place for the comparison
    opt_g, opt_d = torch.optim.SGD(G.parameters(), lr=0.0), torch.optim.SGD(D.parameters(), lr=0.0)
    clips = S.make_clips(B, 256, 256, tag=cfg["tag"])
    rgb_x, op_x, rgb_t, op_t = (t.to(DEV) for t in clips)
    rgb = torch.cat([rgb_x.view(B, 4, 3, 256, 256), rgb_t[:, None]], 1)
    op = torch.cat([op_x.view(B, 3, 2, 256, 256), op_t[:, None]], 1)
    g_loss, d_loss = Hn.train_step_gan(G, D, opt_g, opt_d, rgb, op, Hn.flownet_flow_fn(F2), **cfg["lams"])
    assert abs(float(g_loss) - float(d["g_loss"])) <= 1e-4 * abs(float(d["g_loss"]))
    assert abs(float(d_loss) - float(d["d_loss"])) <= 1e-4 * abs(float(d["d_loss"]))
    g_hip = {"G." + n: p.grad.detach().clone() for n, p in G.named_parameters()}
    g_hip.update({"D." + n: p.grad.detach().clone() for n, p in D.named_parameters()})
    idx_hip = T.hip_lookups(G)
    del G, D, F2, opt_g, opt_d
    torch.cuda.empty_cache()
    smp = {n: d[("ggs4k." if n[0] == "G" else "dgs4k.") + n[2:]] for n in g_hip}
    nrm = {n: float(d[("ggn." if n[0] == "G" else "dgn.") + n[2:]]) for n in g_hip}
    return T.same_branch_verdict(T.gan_stepper(sd_g, sd_d, sd_f, clips, cfg["lams"]), g_hip, idx_hip, DEV,
                                 "timed_batch" if B >= 16 else "small_batch", ref=(smp, nrm, T.fixture_idx(d)),
                                 what=f"joint G / D iteration, batch {B}, FlowNet2-SD {flow_precision}")


@pytest.mark.parametrize("flow_precision", ["s16", "fp32"])
def test_gan_iteration_256_vs_reference_vectors(flow_precision):
    T.assert_ok(_iteration(2, flow_precision))


def test_gan_iteration_256_batch32_gradients_against_the_fp64_truth():
    """the batch bench.py's `train_gan` leg times (BASELINE.json configs[2]): every gradient of G and D within 1e-3 (norm)
    of the same-branch truth and, entry by entry, within max(1e-3, 2 x the reference's own error) per tensor"""
    T.assert_ok(_iteration(32, "s16"))
