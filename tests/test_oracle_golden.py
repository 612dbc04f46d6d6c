"""The CPU oracle (oracle/ammc_oracle.py) against vectors recorded from the
reference itself (tests/golden/make_golden.py), and - when /root/reference is
present - against the live reference module.  CPU only."""
import importlib.util
import json
import os
import sys
import types
import warnings

import numpy as np
import pytest
import torch

from ammcnet_aaai2021_amd import synthetic as S
from oracle import ammc_oracle as O
from conftest import GOLDEN, rel_err

warnings.filterwarnings("ignore")
TOL = 1e-6          # oracle vs reference-recorded vectors (same ATen ops -> normally 0)


def _load(name):
    d = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    return d, json.loads(str(d["cfg"])) if "cfg" in d.files else None


def _sub(t, want_shape):
    step = t.shape[-1] // want_shape[-1]
    return t[..., ::step, ::step]


def test_param_counts_known_answers():
    """the only known answers in the reference source: unet.py:1232-1235, 1268-1275"""
    with open(os.path.join(GOLDEN, "param_counts.json")) as fp:
        pc = json.load(fp)
    assert pc["twostream"] == 25049029 and pc["unetmem_v7_rgb"] == 7805891
    sd = S.make_twostream_state()
    assert list(sd.keys()) == pc["twostream_state_keys"]
    leafs = ("running_mean", "running_var", "num_batches_tracked", "embed", "cluster_size", "embed_avg")
    n = sum(v.numel() for k, v in sd.items() if k.rsplit(".", 1)[-1] not in leafs)
    assert n == 25049029
    n_rgb = sum(v.numel() for k, v in sd.items()
                if k.startswith("rgb.") and k.rsplit(".", 1)[-1] not in leafs)
    assert n_rgb == 7805891


@pytest.mark.parametrize("name", ["twostream_64_b2_eval", "twostream_64_b2_m2000_eval", "twostream_256_b2_eval",
                                  "twostream_256_b16_m2000_eval"])
def test_twostream_eval_golden(name):
    d, cfg = _load(name)
    sd = S.make_twostream_state(tuple(cfg["in_channel"]), tuple(cfg["out_channel"]), cfg["embed_dim"],
                                cfg["n_embed"], cfg["k"])
    rgb_x, op_x, rgb_t, _ = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])
    with torch.no_grad():
        rgb, op, (rd, od), (rq, oq), aux = O.twostream_forward(sd, rgb_x, op_x, cfg["k"], want_aux=True)
    step = int(d["out_step"])
    assert rel_err(rgb[..., ::step, ::step], d["rgb"]) <= TOL
    assert rel_err(op[..., ::step, ::step], d["op"]) <= TOL
    assert rel_err(rd, d["rgb_diff"]) <= TOL and rel_err(od, d["op_diff"]) <= TOL
    qs = int(d["q_step"]) if "q_step" in d.files else 1          # the batch-16 fixture stores strided maps / a few rows
    rows = list(d["st_rows"]) if "st_rows" in d.files else slice(None)
    assert rel_err(rq[:, ::qs, ::qs], d["rgb_q"]) <= TOL and rel_err(oq[:, ::qs, ::qs], d["op_q"]) <= TOL
    names = {"rgb.inc": "rgb.x1", "rgb.down1": "rgb.x2", "rgb.down2": "rgb.x3", "rgb.down3": "rgb.x4",
             "rgb.vq_down3": "rgb.vq", "rgb.up1": "rgb.u1", "rgb.up2": "rgb.u2", "rgb.up3": "rgb.u3"}
    for ref_name, mine in list(names.items()):
        names[ref_name.replace("rgb", "op")] = mine.replace("rgb", "op")
    names["rgb.bridge"], names["op.bridge"] = "rgb.bridge", "op.bridge"
    for ref_name, mine in names.items():
        want = d[f"st.{ref_name}"]
        assert rel_err(_sub(aux[mine][rows], want.shape), want) <= TOL, ref_name
    psnr = torch.stack([O.psnr_error(rgb[i:i + 1], rgb_t[i:i + 1]) for i in range(rgb.shape[0])])
    assert rel_err(psnr, d["rgb_psnr"]) <= 1e-6


def test_unet_eval_golden():
    """config 1 of BASELINE.json: plain `UNet(12,3)`, batch 2, CPU"""
    d, cfg = _load("unet_64_b2_eval")
    x = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])[0]
    with torch.no_grad():
        y = O.unet_forward(S.make_unet_state(12, 3), x)
    assert rel_err(y, d["y"]) <= TOL


def test_quantize_cases_golden():
    d, _ = _load("quantize_cases")
    name = "quantize_cases"
    for cname in ("m256", "m2000", "d512m8192", "k3"):
        c = json.loads(str(d[f"{cname}.cfg"]))
        embed = S.hashed_normal(f"{name}:{cname}:embed", (c["d"], c["m"]), 0.9)
        x = S.hashed_normal(f"{name}:{cname}:x", (*c["bhw"], c["d"]), 0.8)
        qk, diff, idxk, idx1, _, q1 = O.quantize_topk(x, embed, c["k"])
        assert np.array_equal(qk.numpy(), d[f"{cname}.qk"]), cname       # gather: bit exact
        assert rel_err(diff, d[f"{cname}.diff"]) <= TOL
        assert np.array_equal((x + (q1 - x)).numpy(), d[f"{cname}.q1"])     # unet.py:311
        assert torch.equal(idxk[..., 0].reshape(-1), idx1)               # topk[:,0] == argmax
    qk, diff, *_ = O.quantize_topk(torch.from_numpy(d["tie.x"]),
                                   S.hashed_normal(f"{name}:tie:embed", (64, 256), 0.9), 2)
    assert np.array_equal(qk.numpy(), d["tie.qk"])
    # one training step: EMA buffers + gradient of the commit term
    sd = {"q.embed": S.hashed_normal(f"{name}:ema:embed", (64, 256), 0.9),
          "q.cluster_size": S.hashed_uniform(f"{name}:ema:cs", (256,), 0.5, 4.0),
          "q.embed_avg": S.hashed_normal(f"{name}:ema:ea", (64, 256), 1.5)}
    x = S.hashed_normal(f"{name}:ema:x", (2, 8, 8, 64), 0.8).requires_grad_(True)
    qk, diff, idxk, idx1, flat, q1 = O.quantize_topk(x, sd["q.embed"], 2)
    O.codebook_ema_update(sd, "q", flat, idx1)
    diff.backward()
    assert np.array_equal(qk.detach().numpy(), d["ema.qk"])
    for key in ("embed", "cluster_size", "embed_avg"):
        assert rel_err(sd[f"q.{key}"], d[f"ema.{key}"]) <= TOL, key
    assert rel_err(x.grad, d["ema.dx"]) <= TOL


def test_twostream_train_step_golden():
    d, cfg = _load("twostream_64_b2_train")
    sd = O.clone_state(S.make_twostream_state(), requires_grad=True)
    rgb_x, op_x, rgb_t, op_t = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])
    out = O.twostream_forward(sd, rgb_x, op_x, cfg["k"], training=True)
    loss = O.generator_loss(out, rgb_t, op_t)
    loss.backward()
    assert rel_err(loss, d["loss"]) <= TOL
    assert rel_err(out[0], d["rgb"]) <= TOL and rel_err(out[1], d["op"]) <= TOL
    for key in d.files:
        if key.startswith("gn."):
            k = key[3:]
            g = sd[k].grad
            assert abs(float(g.double().norm()) - float(d[key])) <= 1e-5 * float(d[key]) + 1e-12, k
            gs = g.flatten()[:: max(1, g.numel() // 64)][:64]
            assert rel_err(gs, d[f"gs.{k}"]) <= 1e-5 or float(np.abs(d[f"gs.{k}"]).max()) < 1e-12, k
        elif key.startswith("buf."):
            k = key[4:]
            assert rel_err(sd[k].double(), d[key].astype(np.float64)) <= TOL, k


def test_eval_records_match_shipped_structure():
    """The record builder reproduces the structure of the authors' own ped2
    pickle: one commit value per batch of 16 clips, first 4 frames back-filled."""
    with open(os.path.join(GOLDEN, "shipped_records_ped2.json")) as fp:
        shipped = json.load(fp)
    assert shipped["keys"] == ["dataset", "op_fea_comm_records", "op_img_pred_records",
                               "rgb_fea_comm_records", "rgb_img_pred_records"]
    counter = {"n": 0}

    def fake_forward(rgb_in, op_in):
        counter["n"] += 1
        b = rgb_in.shape[0]
        v = torch.full((1,), float(counter["n"]))
        return rgb_in[:, :3] * 0.5, op_in[:, :2] * 0.5, (v, v + 0.5), (None, None)

    for vid in shipped["videos"][:3] + shipped["videos"][8:9]:
        t = vid["frames"]
        rgb = S.hashed_uniform(f"vid{t}", (t, 3, 8, 8))
        op = S.hashed_uniform(f"vidop{t}", (t - 1, 2, 8, 8))
        rec = O.eval_subvideo_records(fake_forward, rgb, op)
        comm = rec["rgb_comm"]
        runs, start = [], 0
        for i in range(1, t + 1):
            if i == t or comm[i] != comm[start]:
                runs.append(i - start)
                start = i
        assert runs == vid["commit_runs"]
        assert np.all(rec["rgb_psnr"][:4] == rec["rgb_psnr"][4]) and vid["psnr_head_equal"]


REF_UNET = "/root/reference/Code/models/unet.py"


@pytest.mark.skipif(not os.path.exists(REF_UNET), reason="reference tree only exists in the authoring container")
def test_oracle_vs_live_reference():
    sys.modules["torchsummaryX"] = types.SimpleNamespace(summary=lambda *a, **k: None)
    spec = importlib.util.spec_from_file_location("ref_unet", REF_UNET)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    sd = S.make_twostream_state(tag="live")
    net = ref.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd, strict=True)
    net.eval()
    rgb_x, op_x, _, _ = S.make_clips(1, 32, 48, tag="live")     # non-square, small
    with torch.no_grad():
        want = net(rgb_x, op_x)
        got = O.twostream_forward(O.clone_state(sd), rgb_x, op_x, 2)
    assert rel_err(got[0], want[0]) <= TOL and rel_err(got[1], want[1]) <= TOL
    assert rel_err(got[2][0], want[2][0]) <= TOL and rel_err(got[3][1], want[3][1]) <= TOL


def test_discriminator_and_losses_golden():
    """oracle restatements of PixelDiscriminator / LSGAN / gradient-difference losses against
    values and autograd gradients recorded from the reference classes (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLDEN, "discriminator_64_b2.npz"))
    name = "discriminator_64_b2"
    sd = S.make_discriminator_state()
    assert list(sd.keys()) == list(g["state_keys"]) and sum(v.numel() for v in sd.values()) == int(g["param_count"])
    _, _, real, _ = S.make_clips(2, 64, 64, tag=name)
    fake = (real + 0.3 * S.hashed_uniform(name + ":fake", tuple(real.shape))).clamp(-1, 1).requires_grad_(True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    d_gen = O.pixel_discriminator(sdg, fake)
    assert rel_err(d_gen.detach().numpy(), g["d_gen"]) < 1e-5
    adv = O.adversarial_loss(d_gen)
    assert abs(adv.item() - float(g["adv"])) < 1e-6 * max(1.0, abs(float(g["adv"])))
    adv.backward()
    assert rel_err(fake.grad.numpy(), g["adv_dfake"]) < 1e-4
    for k, v in sdg.items():
        gr = v.grad
        ref = g["adv_dW:" + k]
        got = gr.numpy() if gr.numel() <= 4096 else gr.flatten()[::97].numpy()
        assert rel_err(got, ref) < 1e-4, k
        assert abs(gr.double().norm().item() - float(g["adv_dWnorm:" + k])) < 1e-4 * float(g["adv_dWnorm:" + k])
    fake.grad = None
    gdl = O.gradient_loss(fake, real)
    assert abs(gdl.item() - float(g["gdl"])) < 1e-6
    gdl.backward()
    assert rel_err(fake.grad.numpy(), g["gdl_dfake"]) < 1e-5
    sdd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    d_real, d_fake = O.pixel_discriminator(sdd, real), O.pixel_discriminator(sdd, fake.detach())
    assert rel_err(d_real.detach().numpy(), g["d_real"]) < 1e-5
    dl = O.discriminate_loss(d_real, d_fake)
    assert abs(dl.item() - float(g["d_loss"])) < 1e-6 * max(1.0, abs(float(g["d_loss"])))
    dl.backward()
    for k, v in sdd.items():
        assert abs(v.grad.double().norm().item() - float(g["dis_dWnorm:" + k])) < 1e-4 * float(g["dis_dWnorm:" + k])
    assert abs(O.flow_loss(fake.detach()[:, :2], real[:, :2]).item() - float(g["flow_loss"])) < 1e-6


def test_flownet2sd_golden():
    """the oracle's FlowNet2-SD restatement against outputs recorded from the reference's class"""
    g = np.load(os.path.join(GOLDEN, "flownet2sd_eval.npz"))
    sd = S.make_flownet2sd_state()
    assert list(sd.keys()) == list(g["state_keys"])
    assert sum(v.numel() for v in sd.values()) == int(g["param_count"]) == 45371666      # FlowNetSD.py:4
    for tag in ("a", "b"):
        shape = tuple(int(v) for v in g["shape_" + tag])
        x = (S.hashed_uniform(f"flownet2sd_eval:{tag}", shape) + 1) * 127.5
        with torch.no_grad():
            y = O.flownet2sd_forward(sd, x)
        assert y.shape == (shape[0], 2, shape[3], shape[4])
        assert rel_err(y.numpy(), g["flow_" + tag]) < 1e-6
