"""Round 6, review item 1: what the same-branch fp64 comparison (tests/truth.py) measures on every training fixture whose
gradient gate used to be an envelope fitted to fp32-vs-fp32 distances (1e-2 max / 2e-3 median).  For each case: the HIP
gradients against the truth on the HIP branch (e, norm), and two fp32 witnesses against the truth on THEIR branches - the
oracle on the host (oneDNN) and the oracle on the device (MIOpen / rocBLAS) - so that gates can be read off data.

    python tools/truth_survey.py [case ...]  -> gpurun_out/truth_survey.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ammcnet_aaai2021_amd as A  # noqa: E402
from ammcnet_aaai2021_amd import harness as Hn, synthetic as S  # noqa: E402
from oracle import ammc_oracle as O  # noqa: E402
import truth as T  # noqa: E402

DEV = "cuda:0"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def summarize(tag, v, t0):
    print(f"== {tag}: ok {v['ok']}  norm {v['grad_norm_rel']}  e {v['grad_l2_rel']}  ref e {v['reference_grad_l2_rel']} "
          f"ref norm {v['reference_grad_norm_rel']} ratio {v['ratio_over_reference']}  [{time.time() - t0:.1f} s]", flush=True)
    for w in v["failing"][:12]:
        print("   FAIL", w)


def g_case(hw, batch, tag, precision, sdtag="ammc"):
    sd = S.make_twostream_state(tag=sdtag)
    h, w = hw if isinstance(hw, tuple) else (hw, hw)
    clips = S.make_clips(batch, h, w, tag=tag)
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.train_precision = precision
    rgb_x, op_x, rgb_t, op_t = (t.to(DEV) for t in clips)
    O.generator_loss(net(rgb_x, op_x), rgb_t, op_t).backward()
    g_hip = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    idx_hip = T.hip_lookups(net)
    del net
    torch.cuda.empty_cache()
    out = {}
    t0 = time.time()
    _, t_hip, idx64_forced, _ = T.g_step(sd, clips, torch.float64, DEV, force_idx=idx_hip)
    _, _, idx64, _ = T.g_step(sd, clips, torch.float64, DEV)
    differ = {p: int((idx_hip[p] != idx64[p]).any(dim=1).sum()) for p in idx64}
    for wname, wdev in (("host_fp32", "cpu"), ("device_fp32", DEV)):
        _, wg, widx, _ = T.g_step(sd, clips, torch.float32, wdev)
        _, t_w, _, _ = T.g_step(sd, clips, torch.float64, DEV, force_idx=widx)
        e_ref, n_ref = T.witness_errors(wg, t_w)
        v = T.verdict(g_hip, t_hip, e_ref, n_ref, what=f"{hw} b{batch} {precision} vs {wname}")
        v["lookups_off_unconstrained_fp64"] = differ
        summarize(v["what"], v, t0)
        out[wname] = v
    return out


def gan_case(batch, flow_precision="s16"):
    d = np.load(os.path.join(GOLDEN, f"gan_256_b{batch}_iteration.npz"))
    cfg = json.loads(str(d["cfg"]))
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
    opt_g, opt_d = torch.optim.SGD(G.parameters(), lr=0.0), torch.optim.SGD(D.parameters(), lr=0.0)
    clips = S.make_clips(batch, 256, 256, tag=cfg["tag"])
    rgb_x, op_x, rgb_t, op_t = (t.to(DEV) for t in clips)
    rgb = torch.cat([rgb_x.view(batch, 4, 3, 256, 256), rgb_t[:, None]], 1)
    op = torch.cat([op_x.view(batch, 3, 2, 256, 256), op_t[:, None]], 1)
    g_loss, d_loss = Hn.train_step_gan(G, D, opt_g, opt_d, rgb, op, Hn.flownet_flow_fn(F2), **cfg["lams"])
    gg = {n: p.grad.detach().clone() for n, p in G.named_parameters()}
    dg = {n: p.grad.detach().clone() for n, p in D.named_parameters()}
    idx_hip = T.hip_lookups(G)
    del G, D, F2
    torch.cuda.empty_cache()
    t0 = time.time()
    ridx = {p: torch.as_tensor(d[f"idx.{p}"].astype(np.int64)) for p in ("rgb", "op")}
    tr = T.gan_step(sd_g, sd_d, sd_f, clips, cfg["lams"], torch.float64, DEV, force_idx=ridx)
    print("truth on the reference's branch: g_loss", tr["g_loss"], "fixture", float(d["g_loss"]), "d_loss", tr["d_loss"], float(d["d_loss"]),
          f"[{time.time() - t0:.1f} s]")
    eg, ng = T.reference_errors({n: d["ggs4k." + n] for n in tr["g"]}, {n: d["ggn." + n] for n in tr["g"]}, tr["g"])
    ed, nd = T.reference_errors({n: d["dgs4k." + n] for n in tr["d"]}, {n: d["dgn." + n] for n in tr["d"]}, tr["d"])
    del tr
    th = T.gan_step(sd_g, sd_d, sd_f, clips, cfg["lams"], torch.float64, DEV, force_idx=idx_hip)
    print("hip g_loss", float(g_loss), "truth", th["g_loss"], "d_loss", float(d_loss), th["d_loss"])
    vg = T.verdict(gg, th["g"], eg, ng, what=f"gan b{batch} G ({flow_precision})")
    vd = T.verdict(dg, th["d"], ed, nd, what=f"gan b{batch} D ({flow_precision})")
    summarize(vg["what"], vg, t0)
    summarize(vd["what"], vd, t0)
    for w in vd["worst"]:
        print("   D worst", w)
    return {"G": vg, "D": vd}


CASES = {
    "64_s16": lambda: g_case(64, 2, "twostream_64_b2_train", "s16"),
    "64_fp32": lambda: g_case(64, 2, "twostream_64_b2_train", "fp32"),
    "128_s16": lambda: g_case(128, 4, "train-128", "s16"),
    "100_s16": lambda: g_case((100, 100), 2, "train-100x100", "s16"),
    "100_fp32": lambda: g_case((100, 100), 2, "train-100x100", "fp32"),
    "256_s16": lambda: g_case(256, 2, "twostream_256_b2_train", "s16"),
    "256_fp32": lambda: g_case(256, 2, "twostream_256_b2_train", "fp32"),
    "ddp64_s16": lambda: g_case(64, 4, "syncclips", "s16", sdtag="sync"),
    "gan2": lambda: gan_case(2),
    "gan32": lambda: gan_case(32),
}

if __name__ == "__main__":
    names = sys.argv[1:] or [c for c in CASES if c != "gan32"]
    res = {}
    for n in names:
        try:
            res[n] = CASES[n]()
        except Exception as e:                                   # a survey: keep going
            import traceback
            traceback.print_exc()
            res[n] = {"error": repr(e)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "truth_survey.json"), "w") as fp:
        json.dump(res, fp, indent=1)
