"""Board power and shader clock while one S16 conv layer runs in a loop (direct evidence for the DVFS regime):

    python tools/micro/power_probe.py [seconds per case]

Reads the amdgpu hwmon files (power1_average / power1_input in uW, freq1_input in Hz, power1_cap) every 20 ms from a
thread while the main thread keeps the launch queue full; falls back to `rocm-smi --showpower --showclocks --json`."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import _lib
from tools.conv_bench_lib import make_desc

lib = _lib.load()
SEC = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1][0] != "-" else 3.0


def hwmon():
    """the hwmon directory of THIS process's GPU: matched by PCI address (a box has several cards; the job sees one)"""
    want = None
    try:
        pr = torch.cuda.get_device_properties(0)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception:
        pass
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        files = {n: os.path.join(d, n) for n in os.listdir(d)}
        pw = files.get("power1_average") or files.get("power1_input")
        if pw:
            bdf = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
            cands.append({"power": pw, "cap": files.get("power1_cap"), "freq": files.get("freq1_input"), "dir": d, "bdf": bdf})
    print("cards:", [(c["bdf"], c["dir"]) for c in cands], "want", want, flush=True)
    for c in cands:
        if want and c["bdf"].lower().startswith(want):
            return c
    return cands[0] if cands else None


def rd(path):
    try:
        with open(path) as fp:
            return float(fp.read().strip())
    except Exception:
        return None


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, timeout=10).stdout.decode()
        return json.loads(out)
    except Exception as e:
        return {"error": str(e)}


H = hwmon()
print("hwmon:", H, flush=True)
if H and H["cap"]:
    print("power cap (W):", (rd(H["cap"]) or 0) / 1e6, flush=True)
print("rocm-smi idle:", json.dumps(smi())[:600], flush=True)


def sample_loop(stop, out):
    while not stop.is_set():
        if H:
            out.append((time.time(), rd(H["power"]), rd(H["freq"]) if H["freq"] else None))
        time.sleep(0.02)


def run_case(name, launch, flops):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        launch(s)
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample_loop, args=(stop, samples))
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    n = 0
    e0.record()
    while time.time() - t0 < SEC:
        for _ in range(50):
            launch(s)
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    mid = samples[len(samples) // 4:]                       # skip the ramp
    pw = [p for _, p, _ in mid if p]
    fq = [f for _, _, f in mid if f]
    print(f"{name:44s} {us:8.1f} us/launch  {flops / us / 1e6:7.1f} TF alg  power {sum(pw) / max(len(pw), 1) / 1e6:7.1f} W "
          f"(max {max(pw, default=0) / 1e6:6.1f})  sclk {sum(fq) / max(len(fq), 1) / 1e6:7.1f} MHz  samples {len(mid)}", flush=True)


def conv_case(B, Hh, W, cin, n, const=False, mf=0):
    d, keep = make_desc(B, Hh, W, cin, n, const=const)
    d.s16_mf = mf                                          # 0 = the dispatch's choice, 1 = 32x32x16, 2 = 16x16x32
    label = C.create_string_buffer(96)
    lib.ammc_conv_gemm_s16_variant(C.byref(d), label, 96)
    fl = 2.0 * B * Hh * W * 9 * cin * n
    run_case(f"{cin}->{n} @{Hh} {label.value.decode()}" + (" CONST" if const else ""), lambda s: lib.ammc_conv_gemm_s16(C.byref(d), s), fl)
    return keep


print("idle:", (rd(H["power"]) or 0) / 1e6 if H else None, "W", flush=True)
keep = []
if "--stress" in sys.argv:                                 # config 5: the fp16 memory-addressing kernel at the bench's size
    from ammcnet_aaai2021_amd import synthetic as S
    from ammcnet_aaai2021_amd.workload import MemoryStress
    ms = MemoryStress(S.hashed_normal("stress:e", (512, 8192), 0.9).to("cuda:0"), 2)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(4321)
    xs = torch.randn(256 * 1024, 512, device="cuda:0", generator=g) * 0.8
    run_case("memory_topk_f16, 262144 rows x 8192 slots x 512", lambda s: ms.run(xs), ms.flops(xs.shape[0]))
    sys.exit(0)
if "--wgrad" in sys.argv:                                  # the 3x3 weight-gradient layers of the training step, batch 32
    from ammcnet_aaai2021_amd._lib import AmmcWgradDesc
    from ammcnet_aaai2021_amd.engine import Act, _ptr
    dev = "cuda:0"
    for (B, Hh, W, cin, n) in [(32, 128, 128, 128, 128), (32, 64, 64, 256, 256), (32, 256, 256, 64, 64), (32, 32, 32, 512, 512)]:
        g = torch.Generator(device=dev).manual_seed(1)
        s0 = torch.cuda.current_stream().cuda_stream
        A32 = Act(torch.zeros(B, Hh + 2, W + 2, cin, device=dev), B, Hh, W, cin, 0, 1)
        A32.interior().copy_(torch.rand(B, Hh, W, cin, device=dev, generator=g) * 2 - 1)
        G32 = Act(torch.zeros(B, Hh + 2, W + 2, n, device=dev), B, Hh, W, n, 0, 1)
        G32.interior().copy_(torch.rand(B, Hh, W, n, device=dev, generator=g) * 2 - 1)
        A16 = Act(torch.empty_like(A32.buf), B, Hh, W, cin, 0, 1)
        G16 = Act(torch.empty_like(G32.buf), B, Hh, W, n, 0, 1)
        _lib.check(lib.ammc_split_rows_f32(_ptr(A32.buf), A32.buf.numel(), _ptr(A16.buf), s0), "split a")
        _lib.check(lib.ammc_split_rows_f32(_ptr(G32.buf), G32.buf.numel(), _ptr(G16.buf), s0), "split g")
        kpad = (9 * cin + 31) // 32 * 32
        dwp = torch.zeros(max(n, 32), kpad, device=dev)
        zeros = torch.zeros(1024, device=dev)
        d = AmmcWgradDesc()
        d.g, d.a, d.dw, d.zeros = G16.pix0(), A16.tap0(), _ptr(dwp), _ptr(zeros)
        d.batch, d.height, d.width, d.n, d.cin, d.ntaps, d.a_step = B, Hh, W, n, cin, 9, 1
        d.g_bs, d.g_rs, d.g_ps = G16.strides
        d.a_bs, d.a_rs, d.a_ps = A16.strides
        run_case(f"wgrad {cin}->{n} @{Hh} b{B}", lambda s: lib.ammc_conv_wgrad_s16(C.byref(d), None, s), 2.0 * B * Hh * W * n * cin * 9)
        del A32, G32, A16, G16
    sys.exit(0)
if "--variants" in sys.argv:                               # forced MFMA shapes; AMMC_TAP_KH from the environment
    for shape in [(16, 128, 128, 128, 128), (16, 128, 128, 64, 128), (16, 256, 256, 64, 64), (16, 64, 64, 256, 256)]:
        for mf in (0, 1, 2):
            conv_case(*shape, mf=mf)
    sys.exit(0)
for shape in [(16, 128, 128, 128, 128), (16, 256, 256, 64, 64), (16, 64, 64, 256, 256), (16, 32, 32, 512, 512)]:
    keep.append(conv_case(*shape))
keep.append(conv_case(16, 128, 128, 128, 128, const=True))
# an HBM-bound pass for comparison: device-to-device copy of 1 GiB
x = torch.empty(1 << 28, device="cuda:0")
y = torch.empty_like(x)
run_case("copy 1 GiB (HBM-bound)", lambda s: y.copy_(x), 0.0)
print("rocm-smi after load:", json.dumps(smi())[:600], flush=True)
