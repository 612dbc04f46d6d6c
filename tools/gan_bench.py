"""full alternating G/D iteration (train_helper.py:296-339): G fwd, 2x FlowNet2-SD fwd (the flow term, unless
`noflow`), 3x D fwd, D bwd + Adam, G bwd (through D) + Adam.  usage: python tools/gan_bench.py [batch] [steps] [noflow]"""
import sys, time, json
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import harness as Hn, synthetic as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
G = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
G.load_state_dict(S.make_twostream_state())
G = G.to(dev).train()
D = A.PixelDiscriminator(3, [128, 256, 512, 512])
D.load_state_dict(S.make_discriminator_state())
D = D.to(dev).train()
flow_fn = None
if "noflow" not in sys.argv:
    F2 = A.FlowNet2SD()
    F2.load_state_dict(S.make_flownet2sd_state())
    flow_fn = Hn.flownet_flow_fn(F2.to(dev).eval())
opt_g = torch.optim.Adam(G.parameters(), lr=2e-4)
opt_d = torch.optim.Adam(D.parameters(), lr=2e-5)
rgb_x, op_x, rgb_t, op_t = (t.to(dev) for t in S.make_clips(B, 256, 256, tag="ganbench"))
rgb = torch.cat([rgb_x.view(B, 4, 3, 256, 256), rgb_t[:, None]], 1)
op = torch.cat([op_x.view(B, 3, 2, 256, 256), op_t[:, None]], 1)
def d_only():
    with torch.no_grad():
        fake = rgb_t * 0.9
    dl = Hn.discriminate_loss(D(rgb_t), D(fake))
    opt_d.zero_grad(set_to_none=True); dl.backward(); opt_d.step()
Hn.train_step_gan(G, D, opt_g, opt_d, rgb, op, flow_fn, **Hn.LAMS_ANOPRED); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    gl, dl = Hn.train_step_gan(G, D, opt_g, opt_d, rgb, op, flow_fn, **Hn.LAMS_ANOPRED)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
d_only(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    d_only()
torch.cuda.synchronize()
dd = (time.perf_counter() - t0) / steps
print(json.dumps({"batch": B, "ms_per_iteration": round(dt * 1e3, 2), "clips_per_s": round(B / dt, 2),
                  "d_update_ms": round(dd * 1e3, 2), "flow_term": flow_fn is not None, "g_loss": float(gl), "d_loss": float(dl),
                  "d_slots": D._engine.slots_created}))
