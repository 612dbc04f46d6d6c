"""G-parameter gradient error (HIP vs oracle fp32, oracle fp32 vs fp64) for a fixed cotangent on the rgb output"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S, harness as Hn
from oracle import ammc_oracle as O
DEV = "cuda:0"
torch.set_num_threads(16)
def l2(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
gsd = S.make_twostream_state()
rgb_x, op_x, rgb_t, op_t = S.make_clips(2, 64, 64, tag="gan-step")
def oracle(dt, cot):
    go = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in O.clone_state(gsd).items()}
    for k, v in go.items():
        leaf = k.rsplit(".", 1)[-1]
        if v.is_floating_point() and leaf not in ("running_mean", "running_var", "embed", "cluster_size", "embed_avg"):
            v.requires_grad_(True)
    want = O.twostream_forward(go, rgb_x.to(dt), op_x.to(dt), 2, training=True)
    (want[0] * cot.to(dt)).sum().backward()
    return {k: v.grad for k, v in go.items() if v.requires_grad and v.grad is not None and float(v.grad.abs().max()) > 0}
for kind in ("sign", "smooth"):
    if kind == "smooth":
        cot = S.hashed_uniform("cot", (2, 3, 8, 8)).repeat_interleave(8, 2).repeat_interleave(8, 3) / 1000
    else:
        cot = torch.sign(S.hashed_uniform("cot2", (2, 3, 64, 64))) / 1000
    G = A.get_twostream((12, 6), (3, 2), 64, 256, 2); G.load_state_dict(gsd); G = G.to(DEV).train()
    out = G(rgb_x.to(DEV), op_x.to(DEV))
    (out[0] * cot.to(DEV)).sum().backward()
    a, b = oracle(torch.float32, cot), oracle(torch.float64, cot)
    e_h = np.array([l2(dict(G.named_parameters())[k].grad, b[k]) for k in b])
    e_o = np.array([l2(a[k], b[k]) for k in b])
    e_ho = np.array([l2(dict(G.named_parameters())[k].grad, a[k]) for k in b])
    print(kind, "HIP vs f64: max %.2e med %.2e | oracle32 vs f64: max %.2e med %.2e | HIP vs oracle32: max %.2e med %.2e"
          % (e_h.max(), np.median(e_h), e_o.max(), np.median(e_o), e_ho.max(), np.median(e_ho)))
P = dict(G.named_parameters())
print("--- per tensor (sign->last run), HIP vs f64 | oracle32 vs f64")
for k in b:
    print("%-40s %.2e %.2e  |g|=%.2e" % (k, l2(P[k].grad, b[k]), l2(a[k], b[k]), float(b[k].norm())))
