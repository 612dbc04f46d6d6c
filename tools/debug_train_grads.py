import sys, json, os
sys.path.insert(0, '.')
import numpy as np, torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
from oracle import ammc_oracle as O
DEV='cuda:0'
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = S.make_twostream_state()
net = A.get_twostream((12, 6), (3, 2), 64, 256, 2); net.load_state_dict(sd); net = net.to(DEV).train()
rgb_x, op_x, rgb_t, op_t = S.make_clips(2, hw, hw, tag="twostream_64_b2_train")
out = net(rgb_x.to(DEV), op_x.to(DEV))
loss = O.generator_loss(out, rgb_t.to(DEV), op_t.to(DEV)); loss.backward()
msd = O.clone_state(sd, requires_grad=True)
want = O.twostream_forward(msd, rgb_x, op_x, 2, training=True)
O.generator_loss(want, rgb_t, op_t).backward()
for name, p in net.named_parameters():
    a, b = p.grad.cpu().double().flatten(), msd[name].grad.double().flatten()
    print(f"{name:45s} l2rel {float((a-b).norm()/b.norm()):.2e}  maxrel {float((a-b).abs().max()/b.abs().max()):.2e}  norm {float(b.norm()):.3e}")
