"""The build's own scoring loop / train step (ammcnet_aaai2021_amd/harness.py) against the
oracle's restatement of the reference loop and the structure of the authors' shipped pickle."""
import json
import os

import numpy as np
import pytest
import torch

from ammcnet_aaai2021_amd import harness as Hn, synthetic as S
from oracle import ammc_oracle as O
from conftest import GOLDEN


def _fake_model():
    n = {"calls": 0}

    def model(rgb_in, op_in):
        n["calls"] += 1
        v = torch.full((1,), float(n["calls"]))
        return rgb_in[:, :3] * 0.5, op_in[:, :2] * 0.25, (v, v + 0.5), (None, None)
    return model


@pytest.mark.parametrize("t", [6, 21, 37, 120])
def test_records_equal_oracle_loop(t):
    rgb = S.hashed_uniform(f"hv{t}", (t, 3, 8, 8))
    op = S.hashed_uniform(f"ho{t}", (t - 1, 2, 8, 8))
    got = Hn.evaluate_subvideo(_fake_model(), rgb, op)
    want = O.eval_subvideo_records(_fake_model(), rgb, op)
    for key in want:
        assert np.allclose(got[key], want[key], rtol=1e-6), key


def test_dataset_records_match_shipped_structure_and_shard_invariance():
    with open(os.path.join(GOLDEN, "shipped_records_ped2.json")) as fp:
        shipped = json.load(fp)
    lens = [v["frames"] for v in shipped["videos"][:4]]
    vids = [(S.hashed_uniform(f"dv{i}", (t, 3, 8, 8)), S.hashed_uniform(f"do{i}", (t - 1, 2, 8, 8)))
            for i, t in enumerate(lens)]

    def model(rgb_in, op_in):              # commit value depends only on the batch's content
        v = rgb_in.mean().reshape(1)
        return rgb_in[:, :3] * 0.5, op_in[:, :2] * 0.25, (v, v * 2), (None, None)

    full = Hn.evaluate_dataset(model, vids, "ped2")
    assert sorted(full.keys()) == shipped["keys"]
    for vid, comm in zip(shipped["videos"], full["rgb_fea_comm_records"]):
        runs, start = [], 0
        for i in range(1, len(comm) + 1):
            if i == len(comm) or comm[i] != comm[start]:
                runs.append(i - start)
                start = i
        assert runs == vid["commit_runs"]
    # sharding by whole batches gives the same records: emulate 3 ranks without a process group
    plan = [(v, s, e) for v, (r, _) in enumerate(vids) for s, e in Hn.subvideo_batches(r.shape[0])]
    merged = {}
    for rank in range(3):
        for i in Hn.parallel.shard_batches(len(plan), rank, 3):
            v, s, e = plan[i]
            merged[i] = Hn.score_batch(model, vids[v][0], vids[v][1], s, e)
    for v, (r, _) in enumerate(vids):
        idx = [i for i, p in enumerate(plan) if p[0] == v]
        rec = Hn.assemble_records(r.shape[0], [plan[i][1:] for i in idx], [merged[i] for i in idx])
        assert np.array_equal(rec["rgb_comm"], full["rgb_fea_comm_records"][v])
        assert np.array_equal(rec["rgb_psnr"], full["rgb_img_pred_records"][v])


def test_generator_loss_equals_oracle():
    out = (S.hashed_uniform("a", (2, 3, 8, 8)), S.hashed_uniform("b", (2, 2, 8, 8)),
           (torch.tensor([0.3]), torch.tensor([0.7])), (None, None))
    rt, ot = S.hashed_uniform("c", (2, 3, 8, 8)), S.hashed_uniform("d", (2, 2, 8, 8))
    assert torch.allclose(Hn.generator_loss(out, rt, ot, lam_lp=2.0, lam_latent=0.5),
                          O.generator_loss(out, rt, ot, lam_lp=2.0, lam_latent=0.5))
