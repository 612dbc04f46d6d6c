"""Batch-32 gradients against the fp64 truth: for every gradient tensor of the timed training step,
    e_hip  = |g_hip - g64| / |g64|   (whole tensor; g64 = the oracle in float64, evaluated here on the device)
    e_ref  = |g_ref - g64| / |g64|   on the 4096 recorded positions (g_ref: the REFERENCE's own fp32 autograd, fixture)
    e_o32  = the oracle's fp32 evaluation, same positions (second fp32 witness)
for both training precisions, plus the norms.  Prints the tensors that break e_hip <= max(1e-3, 1.5 e_ref).

    python tools/grad_truth.py [--fp64-fixture tests/golden/twostream_256_b32_train_fp64.npz] [--batch 32]"""
import argparse
import json
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests/golden')
import numpy as np
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
from ammcnet_aaai2021_amd import harness as Hn
from make_fp64_truth import dense_samples, oracle_step

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--fp64-fixture", default="")
ap.add_argument("--precisions", default="s16,fp32")
ap.add_argument("--json", default="")
args = ap.parse_args()
d = np.load(f"tests/golden/twostream_256_b{args.batch}_train.npz")
cfg = json.loads(str(d["cfg"]))
dev = "cuda:0"
sd = S.make_twostream_state()
clips_cpu = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])


def l2rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


hip = {}
for prec in args.precisions.split(","):
    clips = [t.to(dev) for t in clips_cpu]
    net = A.get_twostream((12, 6), (3, 2), 64, cfg["n_embed"], cfg["k"])
    net.load_state_dict(sd)
    net = net.to(dev).train()
    net.train_precision = prec
    out = net(clips[0], clips[1])
    loss = Hn.generator_loss(out, clips[2], clips[3])
    loss.backward()
    torch.cuda.synchronize()
    hip[prec] = (float(loss.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
    del net, out, loss, clips
    torch.cuda.empty_cache()

import time
t0 = time.time()
loss64, g64 = oracle_step(sd, clips_cpu, torch.float64, dev)
torch.cuda.synchronize()
print(f"oracle fp64 on the device: {time.time() - t0:.1f} s, loss {loss64!r}")
fx = np.load(args.fp64_fixture) if args.fp64_fixture and os.path.exists(args.fp64_fixture) else None
if fx is not None:
    w = max(l2rel(dense_samples(g64[k]).cpu(), torch.as_tensor(fx[f"gs64.{k}"])) for k in g64)
    print(f"device fp64 vs the committed host fp64 fixture, worst tensor (4096 samples): {w:.2e}; loss {abs(loss64 - float(fx['loss64'])) / loss64:.1e}")
have4k = any(k.startswith("gs4k.") for k in d.files)
report = {}
for prec, (loss, g) in hip.items():
    rows = []
    for n in g:
        t = g64[n]
        s64 = dense_samples(t).cpu()
        e_full = l2rel(g[n], t)
        e_s = l2rel(dense_samples(g[n]).cpu(), s64)
        e_ref = l2rel(torch.as_tensor(d[f"gs4k.{n}"]), s64) if have4k else float("nan")
        e_o32 = l2rel(torch.as_tensor(fx[f"gs32.{n}"]), s64) if fx is not None else float("nan")
        n64 = float(t.norm())
        nh = abs(float(g[n].double().norm()) - n64) / n64
        nr = abs(float(d[f"gn.{n}"]) - n64) / n64
        rows.append((n, e_full, e_s, e_ref, e_o32, nh, nr, t.numel()))
    bad = [r for r in rows if r[1] > max(1e-3, 1.5 * r[3]) or r[5] > max(1e-3, 1.5 * r[6])]
    print(f"\n== {prec}: loss rel {abs(loss - loss64) / loss64:.2e} (reference fixture: {abs(float(d['loss']) - loss64) / loss64:.2e})")
    print(f"   tensors {len(rows)}, breaking e_hip <= max(1e-3, 1.5 e_ref) or the same relation on norms: {len(bad)}")
    for key, idx in (("e_hip_full", 1), ("e_hip_samples", 2), ("e_ref", 3), ("e_oracle32", 4), ("norm_hip", 5), ("norm_ref", 6)):
        v = sorted(r[idx] for r in rows)
        print(f"   {key:14s} max {v[-1]:.2e}  p90 {v[int(0.9 * len(v))]:.2e}  median {v[len(v) // 2]:.2e}")
    print("   name  e_hip_full  e_hip_samples  e_ref  e_oracle32 | norm_hip norm_ref | numel")
    for r in sorted(rows, key=lambda r: -r[1] / max(1e-3, 1.5 * r[3]))[:25]:
        print("   %-44s %.2e %.2e %.2e %.2e | %.2e %.2e | %d" % r)
    report[prec] = [dict(zip(("name", "e_hip_full", "e_hip_samples", "e_ref", "e_oracle32", "norm_hip", "norm_ref", "numel"), r)) for r in rows]
if args.json:
    os.makedirs(os.path.dirname(args.json) or ".", exist_ok=True)
    json.dump(report, open(args.json, "w"), indent=0)
