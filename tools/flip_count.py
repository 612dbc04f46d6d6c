"""Where do two fp32-accurate evaluations of the batch-32 training step part from the fp64 truth?  For every 3x3 conv + BN
unit: the raw convolution output of the HIP engine (split-fp16 and exact-fp32 kernels) and of the oracle in fp32 against
the oracle in fp64 - L2 error of the tensor, and the number of ReLU masks that come out differently (mask = sign of the
BatchNorm of each evaluation's OWN convolution output, computed in fp64) - plus the memory lookups that pick another slot.
Discontinuities (masks, pool routes, lookups) are what moves gradient entries by 1e-2 between evaluations whose forward
values agree to 1e-6 (tools/grad_truth.py).

    python tools/flip_count.py [--batch 32]"""
import argparse
import json
import sys
import types
sys.path.insert(0, '.')
import numpy as np
import torch
import torch.nn.functional as TF
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
from ammcnet_aaai2021_amd import harness as Hn
from oracle import ammc_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
d = np.load(f"tests/golden/twostream_256_b{args.batch}_train.npz")
cfg = json.loads(str(d["cfg"]))
dev = "cuda:0"
sd = S.make_twostream_state()
clips_cpu = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])


def unit_names(prefix):
    out = []
    for blk in ("inc.conv.conv", "down1.mpconv.1.conv", "down2.mpconv.1.conv", "down3.mpconv.1.conv",
                "up1.conv.conv", "up2.conv.conv", "up3.conv.conv"):
        out += [f"{prefix}.{blk}.0", f"{prefix}.{blk}.3"]
    return out


def oracle_raw(dtype):
    """conv outputs (pre-BatchNorm) of every 3x3 unit + lookup indices of one oracle training forward"""
    m = {k: (v.to(device=dev, dtype=dtype) if v.is_floating_point() else v.to(dev)) for k, v in sd.items()}
    byid = {id(v): k for k, v in m.items()}
    raw = {}

    class Proxy(types.ModuleType):
        def __getattr__(self, name):
            return getattr(TF, name)

        @staticmethod
        def conv2d(x, w, *a, **kw):
            y = TF.conv2d(x, w, *a, **kw)
            key = byid.get(id(w), "")
            if w.shape[-1] == 3 and "outc" not in key:
                raw[key[:-len(".weight")]] = y.detach()
            return y
    keep = O.F
    O.F = Proxy("proxyF")
    try:
        with torch.no_grad():
            out = O.twostream_forward(m, clips_cpu[0].to(dev, dtype), clips_cpu[1].to(dev, dtype), 2, training=True, want_aux=True)
    finally:
        O.F = keep
    aux = out[-1]
    return raw, {"rgb": aux["rgb.idx"].reshape(-1, 2), "op": aux["op.idx"].reshape(-1, 2)}


def hip_raw(prec):
    net = A.get_twostream((12, 6), (3, 2), 64, cfg["n_embed"], cfg["k"])
    net.load_state_dict(sd)
    net = net.to(dev).train()
    net.train_precision = prec
    clips = [t.to(dev) for t in clips_cpu]
    out = net(clips[0], clips[1])
    torch.cuda.synchronize()
    st = net._train_engine._last
    raw, idx = {}, {}
    for si, p in enumerate(("rgb", "op")):
        s = st["streams"][si]
        blocks = [s.inc] + list(s.down) + list(s.up_dc)
        for name2, blk in zip(unit_names(p)[::2], blocks):
            raw[name2] = blk.u0.craw.interior().permute(0, 3, 1, 2).clone()
            raw[name2[:-1] + "3"] = blk.u1.craw.interior().permute(0, 3, 1, 2).clone()
        idx[p] = s.idx.reshape(-1, 2).long().clone()
    for key, blk in (("bridge.O2F.conv", st["o2f"]), ("bridge.F20.conv", st["f2o"])):
        raw[key + ".0"] = blk.u0.craw.interior().permute(0, 3, 1, 2).clone()
        raw[key + ".3"] = blk.u1.craw.interior().permute(0, 3, 1, 2).clone()
    del net, out
    torch.cuda.empty_cache()
    return raw, idx


def mask_of(c, key):
    """sign of BatchNorm(c) with c's own batch statistics, in fp64"""
    c = c.double()
    bn = key[:-1] + ("1" if key.endswith("0") else "4")
    g, b = sd[bn + ".weight"].to(dev).double(), sd[bn + ".bias"].to(dev).double()
    mean = c.mean(dim=(0, 2, 3), keepdim=True)
    var = c.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    return ((c - mean) / torch.sqrt(var + 1e-5) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)) > 0


r64, i64 = oracle_raw(torch.float64)
evals = {"oracle32": oracle_raw(torch.float32), "hip_s16": hip_raw("s16"), "hip_fp32": hip_raw("fp32")}
print("unit                               numel      | " + " | ".join(f"{e:>8s}: L2err   flips" for e in evals))
tot = {e: [0, 0.0] for e in evals}
order = unit_names("rgb")[:8] + unit_names("op")[:8] + ["bridge.O2F.conv.0", "bridge.O2F.conv.3", "bridge.F20.conv.0", "bridge.F20.conv.3"] \
    + unit_names("rgb")[8:] + unit_names("op")[8:]
for key in order:
    c64 = r64[key]
    m64 = mask_of(c64, key)
    row = f"{key:34s} {c64.numel():10d} | "
    for e, (raw, _) in evals.items():
        c = raw[key]
        err = float((c.double() - c64).norm() / c64.norm())
        flips = int((mask_of(c, key) != m64).sum())
        tot[e][0] += flips
        row += f"          {err:.2e} {flips:7d} | "
    print(row)
print("flips in all units:", {e: v[0] for e, v in tot.items()})
for p in ("rgb", "op"):
    print(f"memory lookups ({p}) that differ from fp64 (top-1 / either of top-2), of {i64[p].shape[0]}:",
          {e: (int((ix[p][:, 0] != i64[p][:, 0]).sum()), int((ix[p] != i64[p]).any(dim=1).sum())) for e, (_, ix) in evals.items()})
