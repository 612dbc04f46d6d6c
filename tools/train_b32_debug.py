"""Where the batch-32 training step and the reference-recorded fixture differ (tests/golden/twostream_256_b32_train.npz):
both training precisions against the fixture and against each other, buffer by buffer and gradient by gradient, plus
the BatchNorm batch variance of the first layers recomputed in float64 from the engine's own raw conv output.

    python tools/train_b32_debug.py [batch]"""
import json
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
from ammcnet_aaai2021_amd import harness as Hn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = np.load(f"tests/golden/twostream_256_b{B}_train.npz")
cfg = json.loads(str(d["cfg"]))
dev = "cuda:0"
clips = [t.to(dev) for t in S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])]
res = {}
for prec in ("s16", "fp32"):
    net = A.get_twostream((12, 6), (3, 2), 64, cfg["n_embed"], cfg["k"])
    net.load_state_dict(S.make_twostream_state())
    net = net.to(dev).train()
    net.train_precision = prec
    out = net(clips[0], clips[1])
    loss = Hn.generator_loss(out, clips[2], clips[3])
    loss.backward()
    torch.cuda.synchronize()
    sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items() if v.is_floating_point()}
    gn = {n: float(p.grad.double().norm()) for n, p in net.named_parameters()}
    gs = {n: p.grad.detach().flatten()[:: max(1, p.grad.numel() // 64)][:64].double().cpu() for n, p in net.named_parameters()}
    res[prec] = (sd, gn, float(loss.detach()), gs)
    if prec == "s16":
        st = net._train_engine._last
        for si, name in ((0, "rgb"), (1, "op")):
            u = st["streams"][si].inc.u0
            c = u.craw.interior().double()
            var = c.var(dim=(0, 1, 2), unbiased=True)
            rv = 0.9 * 1.0 + 0.1 * var
            key = f"{name}.inc.conv.conv.1.running_var"
            print(f"{key}: fp64 recomputation from the engine's raw conv output vs engine {float((sd[key] - rv.cpu()).abs().max() / rv.abs().max()):.2e}"
                  f"  vs fixture {float((torch.as_tensor(d['buf.' + key]).double() - rv.cpu()).abs().max() / rv.abs().max()):.2e}")
    del net, out, loss
    torch.cuda.empty_cache()


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


print("loss: s16 %.9f fp32 %.9f fixture %.9f" % (res["s16"][2], res["fp32"][2], float(d["loss"])))
rows = []
for k in d.files:
    if k.startswith("buf.") and k[4:] in res["s16"][0]:
        w = torch.as_tensor(np.asarray(d[k])).double()
        rows.append((rel(res["s16"][0][k[4:]], w), rel(res["fp32"][0][k[4:]], w), rel(res["s16"][0][k[4:]], res["fp32"][0][k[4:]]), k[4:]))
rows.sort(reverse=True)
print("buffers: s16-vs-fixture  fp32-vs-fixture  s16-vs-fp32")
for r in rows[:12]:
    print("  %.2e  %.2e  %.2e  %s" % r)
rows = []
for n, g in res["s16"][1].items():
    w = float(d[f"gn.{n}"])
    rows.append((abs(g - w) / w, abs(res["fp32"][1][n] - w) / w, abs(g - res["fp32"][1][n]) / w, n))
rows.sort(reverse=True)
print("gradient norms: s16-vs-fixture  fp32-vs-fixture  s16-vs-fp32")
for r in rows[:12]:
    print("  %.2e  %.2e  %.2e  %s" % r)


def l2rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


for prec in ("s16", "fp32"):
    e = sorted(l2rel(res[prec][3][n], torch.as_tensor(d[f"gs.{n}"]).double()) for n in res[prec][3])
    print(f"64-sample L2 per gradient tensor, {prec} vs fixture: max {e[-1]:.2e}  p90 {e[int(0.9 * len(e))]:.2e}  median {e[len(e) // 2]:.2e}")
e = sorted(l2rel(res["s16"][3][n], res["fp32"][3][n]) for n in res["s16"][3])
print(f"64-sample L2 per gradient tensor, s16 vs fp32 engine: max {e[-1]:.2e}  p90 {e[int(0.9 * len(e))]:.2e}  median {e[len(e) // 2]:.2e}")
for stream in ("rgb", "op"):
    k = f"{stream}.vq_down3.quan.quantize.cluster_size"
    w = torch.as_tensor(np.asarray(d["buf." + k])).double()
    for prec in ("s16", "fp32"):
        moved = float((res[prec][0][k] - w).abs().sum()) / 0.01 / 2          # rows whose nearest slot differs (decay 0.99)
        print(f"{k} {prec}: {moved:.2f} re-routed rows of {B * 32 * 32}")
