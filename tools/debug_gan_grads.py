"""diagnose the G-side gradient of the GAN step term by term (HIP vs oracle on the SAME generated frame)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S, harness as Hn
from oracle import ammc_oracle as O
DEV = "cuda:0"
def l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
gsd = S.make_twostream_state()
G = A.get_twostream((12, 6), (3, 2), 64, 256, 2); G.load_state_dict(gsd); G = G.to(DEV).train()
dsd = S.make_discriminator_state()
D = A.PixelDiscriminator(3, [128, 256, 512, 512]); D.load_state_dict(dsd); D = D.to(DEV).train()
rgb_x, op_x, rgb_t, op_t = S.make_clips(2, 64, 64, tag="gan-step")
out = G(rgb_x.to(DEV), op_x.to(DEV))
x = out[0].detach().clone()
for name, fn_h, fn_o in [
    ("adv", lambda t: Hn.adversarial_loss(D(t)), lambda t: O.adversarial_loss(O.pixel_discriminator(dsd, t))),
    ("gdl", lambda t: Hn.gradient_loss(t, rgb_t.to(DEV)), lambda t: O.gradient_loss(t, rgb_t)),
    ("l2", lambda t: torch.norm(t - rgb_t.to(DEV), p=2, dim=1).mean(), lambda t: O.intensity_l2(t, rgb_t))]:
    xh = x.clone().requires_grad_(True); lh = fn_h(xh); lh.backward()
    xo = x.cpu().clone().requires_grad_(True); lo = fn_o(xo); lo.backward()
    print(name, "loss", lh.item(), lo.item(), "grad err", l2(xh.grad, xo.grad), "norm", float(xo.grad.norm()))
# full: oracle gradients w.r.t. G params given the HIP d_rgb vs oracle d_rgb
go = O.clone_state(gsd, requires_grad=True)
want = O.twostream_forward(go, rgb_x, op_x, 2, training=True)
print("fwd rgb err", l2(out[0], want[0]), "op", l2(out[1], want[1]))
