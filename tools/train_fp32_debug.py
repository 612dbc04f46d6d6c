"""where do the exact-fp32 and the S16 training backward passes part?  Compares named gradient buffers of the two
engines on the mask-free fixture (the S16 one matches the fp64 oracle to 4e-6 there)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
from oracle import ammc_oracle as O
from test_gpu_train import _mask_free_state

sd = _mask_free_state()
clips = [t.cuda() for t in S.make_clips(2, 64, 64, tag="maskfree")]
eng = {}
for prec in ("s16", "fp32"):
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd); net = net.cuda().train(); net.train_precision = prec
    out = net(clips[0], clips[1])
    O.generator_loss(out, clips[2], clips[3]).backward()
    torch.cuda.synchronize()
    eng[prec] = net._train_engine._last
def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))
for si, sname in enumerate(("rgb", "op")):
    a, b = eng["fp32"]["streams"][si], eng["s16"]["streams"][si]
    items = [("dpre", a.dpre, b.dpre)]
    for j in range(3):
        items.append((f"du[{j}]", a.du[j], b.du[j]))
        items.append((f"dcat[{j}]", a.dcat[j], b.dcat[j]))
        items.append((f"dskip_tot[{j}]", a.dskip_tot[j], b.dskip_tot[j]))
        items.append((f"dpooled[{j}]", a.dpooled[j], b.dpooled[j]))
        items.append((f"up_dc[{j}].dmid", a.up_dc[j].dmid, b.up_dc[j].dmid))
        items.append((f"down[{j}].dmid", a.down[j].dmid, b.down[j].dmid))
        items.append((f"down[{j}].u1.dc", a.down[j].u1.dc, b.down[j].u1.dc))
        items.append((f"down[{j}].u0.dc", a.down[j].u0.dc, b.down[j].u0.dc))
    items += [("dbottom", a.dbottom, b.dbottom), ("dx4", a.dx4, b.dx4), ("inc.dmid", a.inc.dmid, b.inc.dmid),
              ("inc.u1.dc", a.inc.u1.dc, b.inc.u1.dc), ("inc.u0.dc", a.inc.u0.dc, b.inc.u0.dc)]
    for name, x, y in items:
        # fp32 `dc` buffers may be unused in s16 mode (the S16 twin is written instead): compare only where both non-zero
        xi, yi = x.interior(), y.interior()
        if float(yi.abs().max()) == 0.0:
            print(f"{sname}.{name:18s} (s16 engine keeps no fp32 copy)")
            continue
        print(f"{sname}.{name:18s} rel {rel(xi, yi):.3e}   max|s16| {float(yi.abs().max()):.3e}")
