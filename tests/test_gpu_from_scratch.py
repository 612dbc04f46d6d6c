"""Day one of `run_train`: the reference's FROM-SCRATCH state through the kernels (round-5 review, missing #2).

`Quantize_topk.__init__` (models/unet.py:277-280) starts with `cluster_size = 0`, `embed_avg = embed.clone()`; with the EMA
update of :298-309 every slot NOT hit so far becomes `embed = 0.99^t e0 / ~1e-5` ~ 1e5 x N(0, 1) - for the first ~150
steps of a real training run most of the codebook is beyond the fp16 range.  Every other fixture of this repo starts from
`cluster_size ~ U(0.5, 4)`.  tests/golden/twostream_64_b2_from_scratch.npz (`make_golden.py from_scratch`) holds what the
reference does from its own init path (`get_twostream` + `weights_init_normal`, utils/utils.py:328-334, the random draws
replaced by the hash filler of the same distributions): three Adam steps (loss, commit terms, lookups, the three EMA
buffers of both memories after every step), then an eval forward (frames, commit scalars, quantised maps, lookups).

Asserted here:
  * the HIP training steps follow that trajectory (EMA buffers with their 3e5-sized entries included);
  * the eval forward from the trained state, in S16 (the default) and in exact fp32, equals the ORACLE's eval forward on
    the same state to 1e-4 and follows the reference-recorded one;
  * what the S16 path does with the out-of-range codebook: `ammc_pack_codebook_s16_guarded` raises its verdict, the
    engine looks THAT memory up with the fp32 kernel (`memory_fp32_routed == 2`), no batch is re-run on the fp32 plans
    (`s16_fallbacks == 0`); and when a gathered row itself is out of range (forced here: a codebook with ONE sane slot,
    so every second neighbour is a 1e5-sized slot) the guarded split of the gathered rows raises the range flag and the
    batch IS re-run on the fp32 plans, equal to the oracle.
"""
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
NAME = "twostream_64_b2_from_scratch"


def _trained():
    d = np.load(os.path.join(GOLDEN, NAME + ".npz"))
    cfg = json.loads(str(d["cfg"]))
    net = A.get_twostream((12, 6), (3, 2), 64, cfg["n_embed"], cfg["k"])
    net.load_state_dict(S.make_from_scratch_state(), strict=True)
    net = net.to(DEV).train()
    opt = torch.optim.Adam(net.parameters(), lr=cfg["lr"])
    log = []
    for t in range(cfg["steps"]):
        rgb_x, op_x, rgb_t, op_t = (v.to(DEV) for v in S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=f"{NAME}:{t}"))
        out = net(rgb_x, op_x)
        loss = O.generator_loss(out, rgb_t, op_t)
        opt.zero_grad()
        loss.backward()
        opt.step()
        st = net._train_engine._last
        rec = {"loss": float(loss.detach()), "rgb_diff": out[2][0].detach().cpu(), "op_diff": out[2][1].detach().cpu()}
        for si, p in enumerate(("rgb", "op")):
            rec[f"idx.{p}"] = st["streams"][si].idx.reshape(-1, 2).cpu().numpy().astype(np.int64)
            q = getattr(net, p).vq_down3.quan.quantize
            for b in ("embed", "cluster_size", "embed_avg"):
                rec[f"{p}.{b}"] = getattr(q, b).detach().cpu().clone()
        log.append(rec)
    return d, cfg, net, log


def test_training_from_the_reference_init_follows_the_reference_trajectory():
    d, cfg, net, log = _trained()
    for t, rec in enumerate(log):
        # step 0 is one forward + backward from identical parameters: the 1e-4 gates of every other fixture.  Later steps
        # start from parameters Adam has moved by lr * g / |g| per entry - an entry whose gradient is fp32 noise moves the
        # other way in another fp32 evaluation - so the trajectories separate slowly: 2e-4 per step taken (the tolerance of
        # `test_adam_steps_track_the_oracle`)
        tol = 1e-4 + 2e-4 * t
        assert abs(rec["loss"] - float(d[f"step{t}.loss"])) <= tol * abs(float(d[f"step{t}.loss"])), (t, rec["loss"], float(d[f"step{t}.loss"]))
        assert rel_err(rec["rgb_diff"], d[f"step{t}.rgb_diff"]) <= tol and rel_err(rec["op_diff"], d[f"step{t}.op_diff"]) <= tol, t
        for p in ("rgb", "op"):
            want = d[f"step{t}.idx.{p}"].astype(np.int64)
            off = int((rec[f"idx.{p}"] != want).any(axis=1).sum())
            assert off <= (0 if t == 0 else 2), (t, p, off)           # lookups: identical from identical parameters
            cs, want_cs = rec[f"{p}.cluster_size"].double(), torch.as_tensor(d[f"step{t}.{p}.cluster_size"]).double()
            assert int((cs == 0).sum()) == int((want_cs == 0).sum()) or t > 0, (t, p)
            assert rel_err(cs, want_cs) <= max(tol, 0.02 * off), (t, p)
            assert rel_err(rec[f"{p}.embed_avg"], d[f"step{t}.{p}.embed_avg"]) <= max(tol, 0.02 * off), (t, p)
            # `embed` = embed_avg / smoothed cluster size: the un-hit slots' ~1e5-sized entries, to fp32 accuracy
            assert rel_err(rec[f"{p}.embed"], d[f"step{t}.{p}.embed"]) <= max(tol, 0.02 * off), (t, p)
    for p in ("rgb", "op"):
        e = getattr(net, p).vq_down3.quan.quantize.embed
        assert float(e.abs().max()) > 65504.0                          # the state the S16 packs have to survive
        assert abs(float(e.abs().max()) - float(d[f"final.{p}.embed_absmax"])) <= 1e-3 * float(d[f"final.{p}.embed_absmax"])


@pytest.mark.parametrize("precision", ["s16", "fp32"])
def test_eval_after_training_from_scratch(precision):
    d, cfg, net, _ = _trained()
    net.eval()
    net.precision = precision
    clips = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=f"{NAME}:eval")
    with torch.no_grad():
        rgb, op, (rd, od), (rq, oq) = net(clips[0].to(DEV), clips[1].to(DEV))
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        w = O.twostream_forward(sd, clips[0], clips[1], cfg["k"], want_aux=True)
    eng = net._last_engine
    if precision == "s16":
        # both codebooks hold un-hit slots beyond the half range: looked up by the fp32 kernel, nothing re-run
        assert eng.precision == "s16" and eng.memory_fp32_routed == 2 and not eng.weights_out_of_range
        assert getattr(net, "s16_fallbacks", 0) == 0
    # against the oracle on the SAME (HIP-trained) state: the gates of every eval fixture
    assert rel_err(rgb.cpu(), w[0]) <= 1e-4 and rel_err(op.cpu(), w[1]) <= 1e-4
    assert rel_err(rd.cpu(), w[2][0]) <= 1e-4 and rel_err(od.cpu(), w[2][1]) <= 1e-4
    assert rel_err(rq.cpu(), w[3][0]) <= 1e-4 and rel_err(oq.cpu(), w[3][1]) <= 1e-4
    for si, p in enumerate(("rgb", "op")):
        got = eng._last["streams"][si].idx.reshape(-1, 2).cpu().long()
        assert torch.equal(got, w[-1][f"{p}.idx"].reshape(-1, 2)), p
    # ... and against what the REFERENCE returned after ITS three steps.  Not a kernel gate (that is the 1e-4 above): the two
    # states are three Adam steps apart, and Adam's first steps move every entry by lr * g / |g| - an entry whose gradient
    # is at fp32 noise moves by 2 lr = 4e-4 the other way in another fp32 evaluation, against filters of size 0.02.
    # Measured 3.9e-3 of max|frame| (S16 and fp32 alike); held to 1e-2 so that a wrong trajectory (a missed EMA update, a
    # stale pack after the optimizer step) cannot pass.
    assert rel_err(rgb.cpu(), d["eval.rgb"]) <= 1e-2 and rel_err(op.cpu(), d["eval.op"]) <= 1e-2
    assert rel_err(rd.cpu(), d["eval.rgb_diff"]) <= 1e-2 and rel_err(od.cpu(), d["eval.op_diff"]) <= 1e-2


def test_a_gathered_row_beyond_the_half_range_raises_the_flag_at_the_split():
    """ONE sane slot per codebook: the second neighbour of every row is a slot at ~1e5.  The fp32-routed lookup gathers
    it, `ammc_split_rows_guarded_f32` raises the plan's range flag where the row is re-encoded for `dec`, the guard
    re-runs the batch on the exact-fp32 plans: equal to the oracle."""
    sd = S.make_twostream_state()
    for p in ("rgb", "op"):
        e = sd[f"{p}.vq_down3.quan.quantize.embed"]
        e[:, 1:] = e[:, 1:] * 2.0e5
        # (the second neighbour's half of `dec` scaled down so that the 1e5-sized rows reach the decoder as O(1) values:
        # with the bottleneck at 1e5 the frames saturate and single pixels flip sign on the last bit of an fp32 sum)
        sd[f"{p}.vq_down3.quan.dec.weight"][:, 64:] *= 1.0e-5
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    clips = S.make_clips(2, 64, 64, tag="one-sane-slot")
    with torch.no_grad():
        rgb, op, (rd, od), _ = net(clips[0].to(DEV), clips[1].to(DEV))
        w = O.twostream_forward(O.clone_state(sd), clips[0], clips[1], 2)
    assert net._engine.memory_fp32_routed == 2 and net.s16_fallbacks == 1 and net._last_engine.precision == "fp32"
    assert rel_err(rgb.cpu(), w[0]) <= 1e-4 and rel_err(op.cpu(), w[1]) <= 1e-4
    assert rel_err(rd.cpu(), w[2][0]) <= 1e-4 and rel_err(od.cpu(), w[2][1]) <= 1e-4


def test_the_guarded_entries_alone():
    from ammcnet_aaai2021_amd import _lib
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    flag = torch.zeros(2, device=DEV, dtype=torch.int32)
    x = torch.linspace(-60000.0, 60000.0, 4096, device=DEV).contiguous()
    y = torch.empty_like(x)
    _lib.check(lib.ammc_split_rows_guarded_f32(x.data_ptr(), x.numel(), y.data_ptr(), flag.data_ptr(), s), "split")
    assert flag.tolist() == [0, 0]
    x[1234] = 7.0e4
    _lib.check(lib.ammc_split_rows_guarded_f32(x.data_ptr(), x.numel(), y.data_ptr(), flag.data_ptr(), s), "split")
    assert flag.tolist() == [1, 0]
    x[1234] = float("nan")
    flag.zero_()
    _lib.check(lib.ammc_split_rows_guarded_f32(x.data_ptr(), x.numel(), y.data_ptr(), flag.data_ptr(), s), "split")
    assert flag.tolist() == [1, 0]
    e = S.hashed_normal("guard:e", (64, 300), 1.0).to(DEV).contiguous()
    out = torch.empty((8, 320, 16), device=DEV, dtype=torch.float16)
    flag.zero_()
    _lib.check(lib.ammc_pack_codebook_s16_guarded(e.data_ptr(), 64, 300, out.data_ptr(), flag.data_ptr() + 4, s), "pack")
    assert flag.tolist() == [0, 0]
    e[17, 299] = -1.0e5
    _lib.check(lib.ammc_pack_codebook_s16_guarded(e.data_ptr(), 64, 300, out.data_ptr(), flag.data_ptr() + 4, s), "pack")
    assert flag.tolist() == [0, 1]
    # the unguarded entries are the guarded ones with a null flag
    _lib.check(lib.ammc_pack_codebook_s16(e.data_ptr(), 64, 300, out.data_ptr(), s), "pack")
    _lib.check(lib.ammc_split_rows_f32(x.data_ptr(), x.numel(), y.data_ptr(), s), "split")
    torch.cuda.synchronize()
