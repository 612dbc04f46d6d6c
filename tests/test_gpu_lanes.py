"""The two concurrency switches that are ON by default - the eval forward's two network streams on two HIP streams
(engine.EVAL_LANES, AMMC_EVAL_LANES) and the training step's rgb / flow halves on two HIP streams (train.TWO_STREAMS,
AMMC_TWO_STREAMS) - against the same launches on ONE stream.  Same kernels, disjoint buffers: the forward outputs and
buffers must be BIT-identical; gradients that end in fp32 atomics are held to the run-to-run noise of one configuration."""
import numpy as np
import pytest
import torch

import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import engine as E, synthetic as S, train as T
from ammcnet_aaai2021_amd import harness as Hn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _eval_once(lanes: bool, hw, batch, n_embed):
    keep = E.EVAL_LANES
    E.EVAL_LANES = lanes
    try:
        net = A.get_twostream((12, 6), (3, 2), 64, n_embed, 2)
        net.load_state_dict(S.make_twostream_state(n_embed=n_embed))
        net = net.to(DEV).eval()
        rgb_x, op_x, _, _ = (t.to(DEV) for t in S.make_clips(batch, hw, hw, tag="lanes"))
        with torch.no_grad():
            outs = [net(rgb_x, op_x) for _ in range(2)]          # twice: the second forward re-uses every workspace buffer
        torch.cuda.synchronize()
        used = getattr(net._engine, "_lanes", None) is not None
        return outs, used
    finally:
        E.EVAL_LANES = keep


@pytest.mark.parametrize("hw,batch,n_embed", [(64, 2, 256), (256, 4, 2000)])
def test_eval_lanes_on_equals_off(hw, batch, n_embed):
    on, used_on = _eval_once(True, hw, batch, n_embed)
    off, used_off = _eval_once(False, hw, batch, n_embed)
    assert used_on and not used_off                               # the switch did switch
    for o in (on, off):                                           # and a forward is repeatable within a configuration
        assert torch.equal(o[0][0], o[1][0]) and torch.equal(o[0][1], o[1][1])
    a, b = on[1], off[1]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                      # predicted frames, both streams
    assert torch.equal(a[2][0], b[2][0]) and torch.equal(a[2][1], b[2][1])          # commit values
    assert torch.equal(a[3][0], b[3][0]) and torch.equal(a[3][1], b[3][1])          # quantised maps


def _train_once(two: bool, hw, batch):
    keep = T.TWO_STREAMS
    T.TWO_STREAMS = two
    try:
        net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
        net.load_state_dict(S.make_twostream_state())
        net = net.to(DEV).train()
        rgb_x, op_x, rgb_t, op_t = (t.to(DEV) for t in S.make_clips(batch, hw, hw, tag="lanes-train"))
        out = net(rgb_x, op_x)
        loss = Hn.generator_loss(out, rgb_t, op_t)
        loss.backward()
        torch.cuda.synchronize()
        side = net._train_engine._last["ops"].side is not None
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        bufs = {k: v.detach().clone() for k, v in net.state_dict().items() if k not in grads}
        return (out[0].detach().clone(), out[1].detach().clone(), float(loss.detach())), grads, bufs, side
    finally:
        T.TWO_STREAMS = keep


@pytest.mark.parametrize("hw,batch", [(64, 2), (256, 2)])
def test_training_two_streams_on_equals_off(hw, batch):
    on = _train_once(True, hw, batch)
    off = _train_once(False, hw, batch)
    off2 = _train_once(False, hw, batch)
    assert on[3] and not off[3]
    assert torch.equal(on[0][0], off[0][0]) and torch.equal(on[0][1], off[0][1]) and on[0][2] == off[0][2]
    for k in off[2]:                                              # BatchNorm statistics, step counters, EMA codebook
        assert torch.equal(on[2][k], off[2][k]), k
    for n in off[1]:
        # (some weight gradients are accumulated with fp32 atomics: their run-to-run noise on ONE stream is the gate)
        noise = float((off[1][n] - off2[1][n]).abs().max())
        assert float((on[1][n] - off[1][n]).abs().max()) <= max(4.0 * noise, 0.0) + 1e-7 * float(off[1][n].abs().max()), n
