"""FlowNet2-SD forward on the HIP kernels against vectors recorded from the reference's FlowNet2SD
class (tests/golden/flownet2sd_eval.npz) and the oracle at another size.  Tolerance 2e-5 of max|flow| (26 fp32 layers,
multi-part accumulation of the concatenated inputs in a different order than ATen's single convolution)."""
import os

import numpy as np
import pytest
import torch

from ammcnet_aaai2021_amd import _lib, synthetic as S
from ammcnet_aaai2021_amd.flownet import FlowNet2SD
from oracle import ammc_oracle as O
from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net():
    sd = S.make_flownet2sd_state()
    net = FlowNet2SD()
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    return net.to(DEV).eval(), sd


@pytest.mark.parametrize("precision", ["s16", "fp32"])
def test_flownet2sd_golden_and_oracle(precision):
    """both arithmetic forms: the split-fp16 kernels (default: S16 activations and filters, fp32 flow heads) and the
    exact-fp32 kernels, same gates"""
    g = np.load(os.path.join(GOLDEN, "flownet2sd_eval.npz"))
    net, sd = _net()
    net.precision = precision
    assert sum(p.numel() for p in net.parameters()) == int(g["param_count"])
    for tag in ("a", "b"):
        shape = tuple(int(v) for v in g["shape_" + tag])
        x = (S.hashed_uniform(f"flownet2sd_eval:{tag}", shape) + 1) * 127.5
        y = net(x.to(DEV))
        assert y.shape == (shape[0], 2, shape[3], shape[4])
        assert rel_err(y.cpu().numpy(), g["flow_" + tag]) <= 2e-5
    x = (S.hashed_uniform("flow-other", (1, 3, 2, 128, 192)) + 1) * 127.5
    want = O.flownet2sd_forward(sd, x)
    assert rel_err(net(x.to(DEV)).cpu(), want) <= 2e-5
    assert net._engine.s16 == (precision == "s16")


def test_flownet2sd_interface():
    net, _ = _net()
    with pytest.raises(_lib.AmmcHipError):
        net(torch.zeros(1, 3, 2, 64, 64))                      # CPU tensor: no fallback
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 2, 96, 64, device=DEV))          # six stride-2 levels
    with pytest.raises(NotImplementedError):
        net.train()(torch.zeros(1, 3, 2, 64, 64, device=DEV))
    with pytest.raises(NotImplementedError):
        FlowNet2SD(batchNorm=True)


def test_s16_range_guard_of_flownet():
    """an activation beyond the half range inside the frozen estimator (here: a first layer scaled by 1e5) sets the sticky
    device flag of the S16 kernels; the forward reads it once, recomputes the batch on the exact-fp32 kernels and counts
    the fallback - the flows equal the fp32 model's, not NaN; "defer" hands the flag to the trainer instead"""
    net, sd = _net()
    sd = {k: v.clone() for k, v in sd.items()}
    sd["conv0.0.weight"] *= 1e5
    net.load_state_dict(sd, strict=True)
    x = (S.hashed_uniform("flow-guard", (1, 3, 2, 64, 128)) + 1) * 127.5
    ref = FlowNet2SD()
    ref.load_state_dict(sd, strict=True)
    ref = ref.to(DEV).eval()
    ref.precision = "fp32"
    want = ref(x.to(DEV))
    assert torch.isfinite(want).all()
    got = net(x.to(DEV))
    assert net.s16_fallbacks == 1 and torch.equal(got, want)             # the same fp32 kernels, the same launches
    net.s16_guard = "defer"
    net(x.to(DEV))
    assert int(net.last_overflow) != 0 and net.s16_fallbacks == 1
    del net.s16_guard
    plain, _ = _net()                                                    # ordinary weights: the flag stays clear
    plain(x.to(DEV))
    assert getattr(plain, "s16_fallbacks", 0) == 0
