"""Board power and shader clock of the batch-32 training step and of its HBM-bound BatchNorm passes run alone
(amdgpu hwmon sampling every 20 ms while the main thread keeps the queue full):

    python tools/micro/train_power.py [seconds per case]

What to read from it: whether the streaming passes (6 TB/s of HBM traffic, no MFMA) leave power headroom under the
1400 W cap that the convolution kernels use up - i.e. whether running the two streams of the network on two HIP
streams (one stream's BatchNorm passes beside the other's convolutions) could shorten an energy-bound step."""
import glob
import os
import sys
import threading
import time
sys.path.insert(0, '.')
import torch
import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import _lib, harness, synthetic as S
from ammcnet_aaai2021_amd.engine import Act, _ptr

SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
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


HW = None


def run_case(name, launch, per_launch_bytes=0):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []

    def loop():
        while not stop.is_set():
            if HW:
                samples.append((rd(os.path.join(HW, "power1_input")), rd(os.path.join(HW, "freq1_input"))))
            time.sleep(0.02)
    th = threading.Thread(target=loop)
    th.start()
    t0, n = time.time(), 0
    while time.time() - t0 < SEC:
        for _ in range(4):
            launch()
        n += 4
        torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    stop.set()
    th.join()
    mid = [s for s in samples[len(samples) // 3:] if s[0]]
    pw = sum(p for p, _ in mid) / max(len(mid), 1) / 1e6
    fq = sum(f for _, f in mid) / max(len(mid), 1) / 1e6
    extra = f"  {per_launch_bytes / dt / 1e12:5.2f} TB/s" if per_launch_bytes else ""
    print(f"  {name:64s} {dt * 1e3:9.3f} ms  {pw:7.1f} W  {fq:7.1f} MHz  {pw * dt:8.3f} J{extra}", flush=True)


def main():
    global HW
    HW = hwmon()
    print(f"hwmon {HW}  cap {(rd(os.path.join(HW, 'power1_cap')) or 0) / 1e6:.0f} W" if HW else "no hwmon", flush=True)
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    # ---- the BatchNorm passes alone, on the tensors of the 256x256 level at batch 32 (537 MB each)
    B, H, W, c = 32, 256, 256, 64
    x = Act(torch.randn(B, H + 2, W + 2, c, device=DEV), B, H, W, c, 0, 1)
    dy = Act(torch.randn(B, H + 2, W + 2, c, device=DEV) * 1e-3, B, H, W, c, 0, 1)
    y16 = Act(torch.empty(B, H + 2, W + 2, c, device=DEV), B, H, W, c, 0, 1)
    scale, shift = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.1
    mean, invstd = torch.randn(c, device=DEV) * 0.1, torch.rand(c, device=DEV) + 0.5
    nblk = lib.ammc_chan_reduce_blocks(B * H * W)
    partial = torch.empty(nblk, 4, c, device=DEV)
    sums = torch.zeros(2 * c, device=DEV)
    amax = torch.zeros(256, dtype=torch.int32, device=DEV)
    amax[0] = 0x3a800000                                     # 2^-10
    inv = torch.empty(8, device=DEV)
    elems = B * H * W * c
    run_case("scale_shift_act_s16 (apply forward: 4 B in, 4 B out per element)",
             lambda: lib.ammc_scale_shift_act_s16_f32(x.pix0(), *x.strides, _ptr(scale), _ptr(shift), None, 0, 0, 0, None, y16.pix0(),
                                                      *y16.strides, 1, B, H, W, c, s), 8 * elems)
    run_case("bn_stats (forward reduction: 4 B in per element)",
             lambda: lib.ammc_bn_stats_f32(x.pix0(), *x.strides, B, H, W, c, _ptr(partial), s), 4 * elems)
    run_case("bn_bwd_reduce_bound (backward reduction: 8 B in per element)",
             lambda: lib.ammc_bn_bwd_reduce_bound_f32(x.pix0(), *x.strides, dy.pix0(), *dy.strides, _ptr(mean), _ptr(invstd), _ptr(scale),
                                                      _ptr(shift), 1, B, H, W, c, _ptr(partial), s), 8 * elems)
    run_case("bn_bwd_apply_s16 (backward apply: 8 B in, 4 B out per element)",
             lambda: lib.ammc_bn_bwd_apply_s16_f32(x.pix0(), *x.strides, dy.pix0(), *dy.strides, _ptr(mean), _ptr(invstd), _ptr(scale),
                                                   _ptr(shift), _ptr(sums), 1, y16.pix0(), None, *y16.strides, B, H, W, c,
                                                   amax.data_ptr(), _ptr(inv), 8, s), 12 * elems)
    del x, dy, y16
    # ---- the whole step
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(S.make_twostream_state())
    net = net.to(DEV).train()
    opt = harness.adam(net.parameters(), lr=1e-4)
    rgb_x, op_x, rgb_t, op_t = (t.to(DEV) for t in S.make_clips(32, 256, 256, tag="trainpower"))
    rgb = torch.cat([rgb_x.view(32, 4, 3, 256, 256), rgb_t[:, None]], 1)
    op = torch.cat([op_x.view(32, 3, 2, 256, 256), op_t[:, None]], 1)
    run_case("training step, batch 32 (forward + backward + Adam)", lambda: harness.train_step(net, opt, rgb, op))


if __name__ == "__main__":
    main()
