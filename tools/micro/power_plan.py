"""Board power and shader clock under EVERY kernel of the inference step, one at a time: each distinct launch of the
batch-16 256x256 forward (the headline workload) is repeated for ~1.5 s with the launch queue kept full while a thread
samples the amdgpu hwmon files of this GPU; then energy per step = sum over kernels of power x time x launches.

    python tools/micro/power_plan.py [seconds per kernel]"""
import glob
import os
import sys
import threading
import time
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import synthetic as S

SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
DEV = "cuda:0"


def hwmon():
    pr = torch.cuda.get_device_properties(0)
    want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        bdf = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
        if bdf.lower().startswith(want) and os.path.exists(os.path.join(d, "power1_input")):
            return d
    return None


def rd(path):
    try:
        with open(path) as fp:
            return float(fp.read().strip())
    except Exception:
        return None


H = hwmon()
net = A.get_twostream((12, 6), (3, 2), 64, 2000, 2)
net.load_state_dict(S.make_twostream_state(n_embed=2000), strict=True)
net = net.to(DEV).eval()
B, HW = 16, 256
rgb_x, op_x, _, _ = S.make_clips(B, HW, HW, tag="power")
rgb_x, op_x = rgb_x.to(DEV), op_x.to(DEV)
with torch.no_grad():
    for _ in range(3):
        net(rgb_x, op_x)
torch.cuda.synchronize()
eng = net._engine
st = eng._last
stream = torch.cuda.current_stream().cuda_stream
calls = []


def launch(fn, args, meta):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args, stream)
    e1.record()
    assert rc == 0, meta
    calls.append((fn, args, meta, e0, e1))


outs = [torch.empty((B, s.sp.cout, HW, HW), device=DEV) for s in st["streams"]]
eng._launch_all(st, B, HW, HW, [rgb_x, op_x], outs, [None, None], [None, None], stream, launch, early_flag=False)
torch.cuda.synchronize()
step_ms = sum(e0.elapsed_time(e1) for *_, e0, e1 in calls)
groups = {}
for fn, args, meta, e0, e1 in calls:
    g = groups.setdefault(meta.get("kernel") or meta["name"], dict(ms=0.0, n=0, rep=None, rep_ms=0.0))
    ms = e0.elapsed_time(e1)
    g["ms"] += ms
    g["n"] += 1
    if ms > g["rep_ms"]:
        g["rep"], g["rep_ms"] = (fn, args), ms                 # the longest launch of the group stands for it
print(f"hwmon {H}  cap {rd(os.path.join(H, 'power1_cap')) / 1e6:.0f} W   idle {rd(os.path.join(H, 'power1_input')) / 1e6:.0f} W   "
      f"step (sum of launches, event-bracketed) {step_ms:.3f} ms", flush=True)
tot_j = 0.0
rows = []
for name, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
    if g["ms"] < 0.004 * step_ms:
        continue
    fn, args = g["rep"]
    stop, samples = threading.Event(), []

    def loop():
        while not stop.is_set():
            samples.append((rd(os.path.join(H, "power1_input")), rd(os.path.join(H, "freq1_input"))))
            time.sleep(0.02)
    th = threading.Thread(target=loop)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < SEC:
        for _ in range(40):
            fn(*args, stream)
        n += 40
        torch.cuda.synchronize()
    us = (time.time() - t0) / n * 1e6
    stop.set()
    th.join()
    mid = samples[len(samples) // 3:]
    pw = sum(p for p, _ in mid) / len(mid) / 1e6
    fq = sum(f for _, f in mid) / len(mid) / 1e6
    joule = pw * g["ms"] * 1e-3                                 # at the in-step duration of the group
    tot_j += joule
    rows.append((name, g["n"], g["ms"], us, pw, fq, joule))
    print(f"{name:36s} x{g['n']:2d}  {g['ms']:6.3f} ms in the step   alone: {us:7.1f} us/launch  {pw:7.1f} W  {fq:7.1f} MHz   "
          f"-> {joule:5.2f} J per step", flush=True)
print(f"sum {tot_j:.2f} J per step over {sum(r[2] for r in rows):.3f} ms  = {tot_j / (sum(r[2] for r in rows) * 1e-3):.0f} W average "
      f"(each kernel's sustained-loop power x its in-step time)", flush=True)
