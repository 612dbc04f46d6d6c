"""fwd+bwd+Adam throughput of the training path (BASELINE.json configs[2]: batch 32, 256x256)."""
import sys, time, json
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
net.load_state_dict(S.make_twostream_state())
net = net.to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
rgb_x, op_x, rgb_t, op_t = (t.to(dev) for t in S.make_clips(B, 256, 256, tag="trainbench"))
def step():
    opt.zero_grad(set_to_none=True)
    rgb, op, (rd, od), _ = net(rgb_x, op_x)
    loss = torch.norm(rgb - rgb_t, p=2, dim=1).mean() + torch.norm(op - op_t, p=2, dim=1).mean() + (rd + od).sum()
    loss.backward()
    opt.step()
    return loss
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
ws = net._train_engine._last["bytes"] / 2**30
print(json.dumps({"batch": B, "ms_per_step": round(dt * 1e3, 2), "clips_per_s": round(B / dt, 2),
                  "tflops_algorithmic": round(B / dt * 504.3e9 / 1e12, 2), "workspace_GiB": round(ws, 2), "loss": float(l)}))
