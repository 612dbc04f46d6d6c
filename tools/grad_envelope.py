"""Per-tensor gradient error of the HIP training step against the oracle in FLOAT64, next to the oracle's own fp32-vs-fp64
noise, for both training precisions:  python tools/grad_envelope.py [maskfree|plain] [batch] [size]"""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
from oracle import ammc_oracle as O
from test_gpu_train import _mask_free_state, _oracle_grads, _l2rel

mode = sys.argv[1] if len(sys.argv) > 1 else "maskfree"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 64
torch.set_num_threads(16)
sd = _mask_free_state() if mode == "maskfree" else S.make_twostream_state()
clips = S.make_clips(B, hw, hw, tag="maskfree" if mode == "maskfree" else "twostream_64_b2_train")
loss64, g64, out64 = _oracle_grads(sd, clips, torch.float64)
_, g32, _ = _oracle_grads(sd, clips, torch.float32)
res = {}
for prec in ("s16", "fp32"):
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.cuda().train()
    net.train_precision = prec
    rgb_x, op_x, rgb_t, op_t = (t.cuda() for t in clips)
    out = net(rgb_x, op_x)
    loss = O.generator_loss(out, rgb_t, op_t)
    loss.backward()
    res[prec] = {n: _l2rel(p.grad.cpu(), g64[n]) for n, p in net.named_parameters()}
    print(prec, "loss rel", abs(float(loss) - loss64) / abs(loss64))
print(f"{'tensor':46s} {'s16':>10s} {'fp32':>10s} {'oracle32':>10s}")
for n in g64:
    e32 = _l2rel(g32[n], g64[n])
    flag = "  <--" if max(res["s16"][n], res["fp32"][n]) > max(1e-4, 3 * e32) else ""
    print(f"{n:46s} {res['s16'][n]:10.2e} {res['fp32'][n]:10.2e} {e32:10.2e}{flag}")
for prec in res:
    e = np.array(list(res[prec].values()))
    print(prec, "max %.2e median %.2e" % (e.max(), np.median(e)))
e = np.array([_l2rel(g32[n], g64[n]) for n in g64])
print("oracle32 max %.2e median %.2e" % (e.max(), np.median(e)))
