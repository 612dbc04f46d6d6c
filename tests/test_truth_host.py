"""tests/truth.py on the host: the gate logic on made-up numbers, and the oracle instrumentation it rests on (forcing an
evaluation's own lookups reproduces that evaluation; the joint-iteration oracle reproduces the reference-recorded losses
of tests/golden/gan_256_b2_iteration.npz is left to the -m gpu suite: 256x256 on 8 host threads is minutes)."""
import torch

from ammcnet_aaai2021_amd import synthetic as S
import truth as T


def _fake(e_hip, e_ref, norm_hip=1e-5, norm_ref=1e-5):
    names = [f"t{i}" for i in range(len(e_hip))]
    base = {n: torch.ones(64, dtype=torch.float64) for n in names}
    # a perturbation orthogonal-ish to the tensor keeps the norm error second order; add the norm error explicitly
    pert = torch.cat([torch.ones(32), -torch.ones(32)]).double()
    g = {n: base[n] * (1 + norm_hip) + e * pert for n, e in zip(names, e_hip)}
    return g, base, ({n: e for n, e in zip(names, e_ref)}, {n: norm_ref for n in names})


def test_timed_batch_gates_are_per_tensor():
    g, t, ref = _fake([5e-4] * 12, [1e-6] * 12)
    assert T.verdict(g, t, [ref], "timed_batch")["ok"]                       # below SURVEY's 1e-3 floor whatever the reference does
    g, t, ref = _fake([3e-3] + [5e-4] * 11, [2e-3] + [1e-6] * 11)
    assert T.verdict(g, t, [ref], "timed_batch")["ok"]                       # within twice the reference's own error
    g, t, ref = _fake([5e-3] + [5e-4] * 11, [2e-3] + [1e-6] * 11)
    v = T.verdict(g, t, [ref], "timed_batch")
    assert not v["ok"] and v["failing"][0].startswith("t0 ")
    g, t, ref = _fake([5e-4] * 12, [1e-6] * 12, norm_hip=2e-3)
    assert not T.verdict(g, t, [ref], "timed_batch")["ok"]                   # a norm off by more than 1e-3
    g, t, ref = _fake([1.9e-3] * 12, [1e-3] * 12)
    assert not T.verdict(g, t, [ref], "timed_batch")["ok"]                   # every tensor inside its gate, the median ratio 1.9 > 1.5


def test_small_batch_gates_are_the_witnesses_envelope():
    g, t, ref = _fake([5e-3] + [4e-4] * 11, [1e-6] + [3e-4] * 10 + [3e-3])   # another tensor of the witness carries the flip
    assert not T.verdict(g, t, [ref], "timed_batch")["ok"]
    assert T.verdict(g, t, [ref], "small_batch")["ok"]
    g, t, ref = _fake([7e-3] + [4e-4] * 11, [1e-6] + [3e-4] * 10 + [3e-3])
    assert not T.verdict(g, t, [ref], "small_batch")["ok"]                   # beyond twice the worst witness tensor
    second = ({n: 4e-3 for n in ref[0]}, ref[1])
    assert T.verdict(g, t, [ref, second], "small_batch")["ok"]               # ... unless a second witness is that noisy


def test_forcing_an_evaluations_own_lookups_reproduces_it():
    sd = S.make_twostream_state()
    clips = S.make_clips(2, 32, 32, tag="truth-host")
    loss, g, idx, _ = T.g_step(sd, clips, torch.float32, "cpu")
    loss2, g2, idx2, _ = T.g_step(sd, clips, torch.float32, "cpu", force_idx=idx)
    assert loss == loss2 and all(torch.equal(idx[p], idx2[p]) for p in ("rgb", "op"))
    assert sorted(idx["pool"]) == ["op.down1", "op.down2", "op.down3", "rgb.down1", "rgb.down2", "rgb.down3"]
    assert all(torch.equal(idx["pool"][k], idx2["pool"][k]) for k in idx["pool"])          # forced pool routes = its own
    assert all(torch.equal(g[n], g2[n]) for n in g)
    # another branch: swap the two picks of one row -> the commit term (top-1) and the gathered pair change
    other = {p: v.clone() for p, v in idx.items() if p != "pool"}
    other["rgb"][0] = other["rgb"][0].flip(0)
    loss3, g3, _, _ = T.g_step(sd, clips, torch.float32, "cpu", force_idx=other)
    assert loss3 != loss
    # ... and another pool route: the gradient of the first layer follows it
    moved = {"pool": {k: v.clone() for k, v in idx["pool"].items()}, "rgb": idx["rgb"], "op": idx["op"]}
    moved["pool"]["rgb.down1"][0, :, 0, :] = (moved["pool"]["rgb.down1"][0, :, 0, :] + 1) % 4       # one row of windows, every channel
    loss4, g4, idx4, _ = T.g_step(sd, clips, torch.float32, "cpu", force_idx=moved)
    assert torch.equal(idx4["pool"]["rgb.down1"], moved["pool"]["rgb.down1"])
    assert loss4 != loss and not torch.equal(g4["rgb.inc.conv.conv.3.weight"], g["rgb.inc.conv.conv.3.weight"])
    # the oracle's fp32 evaluation as the evaluation under test: its own reference -> ratio 1, verdict ok
    step = T.g_stepper(sd, clips)
    t64, _ = step(torch.float64, "cpu", idx)
    e, nrm = T.witness_errors(g, t64)
    v = T.verdict(g, t64, [(e, nrm)], "small_batch")
    assert v["ok"] and v["grad_l2_rel"]["max"] == max(e.values())
