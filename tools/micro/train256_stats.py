"""what the 256x256 training fixture (reference-recorded) measures on this build: loss / outputs / buffers / gradient norms and samples"""
import json, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
from oracle import ammc_oracle as O
d = np.load("tests/golden/twostream_256_b2_train.npz")
cfg = json.loads(str(d["cfg"]))
def rel(a, b):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for prec in ("s16", "fp32"):
    sd = S.make_twostream_state()
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.cuda().train()
    net.train_precision = prec
    rgb_x, op_x, rgb_t, op_t = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])
    out = net(rgb_x.cuda(), op_x.cuda())
    loss = O.generator_loss(out, rgb_t.cuda(), op_t.cuda())
    loss.backward()
    st = int(d["out_step"])
    print(prec, "loss rel", abs(float(loss) - float(d["loss"])) / float(d["loss"]),
          "rgb", rel(out[0].detach().cpu()[..., ::st, ::st], d["rgb"]), "op", rel(out[1].detach().cpu()[..., ::st, ::st], d["op"]),
          "diff", rel(out[2][0].detach().cpu(), d["rgb_diff"]), rel(out[2][1].detach().cpu(), d["op_diff"]))
    gn_err, gs_err = [], []
    for name, p in net.named_parameters():
        g = p.grad.detach().cpu()
        gn = float(d[f"gn.{name}"])
        gn_err.append((abs(float(g.double().norm()) - gn) / (gn + 1e-30), name))
        smp = g.flatten()[:: max(1, g.numel() // 64)][:64].double()
        want = torch.as_tensor(d[f"gs.{name}"]).double()
        gs_err.append((float((smp - want).norm() / want.norm().clamp_min(1e-30)), name))
    gn_err.sort(); gs_err.sort()
    print("  grad norm rel: median %.3g max %.3g (%s)" % (gn_err[len(gn_err) // 2][0], gn_err[-1][0], gn_err[-1][1]))
    print("  grad sample L2 rel: median %.3g p90 %.3g max %.3g (%s)" % (gs_err[len(gs_err) // 2][0], gs_err[int(len(gs_err) * 0.9)][0], gs_err[-1][0], gs_err[-1][1]))
    nsd = net.state_dict()
    be = sorted((rel(nsd[k[4:]].cpu(), d[k]), k) for k in d.files if k.startswith("buf."))
    print("  buffers: median %.3g max %.3g (%s)" % (be[len(be) // 2][0], be[-1][0], be[-1][1]))
