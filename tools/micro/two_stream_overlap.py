"""Can an HBM-bound BatchNorm pass hide beside an MFMA-bound convolution of the OTHER stream of the network?

    python tools/micro/two_stream_overlap.py

NC launches of the 128 -> 128 halo-patch convolution at 128x128 (batch 32, fp32 output: a training forward) and NB
launches of the BatchNorm-backward apply pass on a 256x256 x 64 tensor, (a) back to back on one HIP stream, (b) the
convolutions on one stream and the passes on another, started together.  Board power sampled through hwmon meanwhile.
(a) - (b) is what running the rgb and the flow stream of the training step on two HIP streams could gain at best for
this mix; the step's real mix is ~75 % MFMA-bound / ~17 % HBM-bound time."""
import ctypes as C
import glob
import os
import sys
import threading
import time
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import _lib
from ammcnet_aaai2021_amd.engine import Act, _ptr
from tools.conv_bench_lib import make_desc

DEV = "cuda:0"
lib = _lib.load()


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


def main():
    hw = hwmon()
    d, keep = make_desc(32, 128, 128, 128, 128)
    d.y_f32, d.act = 1, 0
    B, H, W, c = 32, 256, 256, 64
    x = Act(torch.randn(B, H + 2, W + 2, c, device=DEV), B, H, W, c, 0, 1)
    dy = Act(torch.randn(B, H + 2, W + 2, c, device=DEV) * 1e-3, B, H, W, c, 0, 1)
    y16 = Act(torch.empty(B, H + 2, W + 2, c, device=DEV), B, H, W, c, 0, 1)
    scale, shift = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.1
    mean, invstd = torch.randn(c, device=DEV) * 0.1, torch.rand(c, device=DEV) + 0.5
    sums = torch.zeros(2 * c, device=DEV)
    amax = torch.zeros(256, dtype=torch.int32, device=DEV)
    amax[0] = 0x3a800000
    inv = torch.empty(8, device=DEV)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def conv(stream):
        _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), stream.cuda_stream), "conv")

    def bn(stream):
        _lib.check(lib.ammc_bn_bwd_apply_s16_f32(x.pix0(), *x.strides, dy.pix0(), *dy.strides, _ptr(mean), _ptr(invstd), _ptr(scale),
                                                 _ptr(shift), _ptr(sums), 1, y16.pix0(), None, *y16.strides, B, H, W, c,
                                                 amax.data_ptr(), _ptr(inv), 8, stream.cuda_stream), "bn")

    def timed(name, fn, reps=6):
        fn()
        torch.cuda.synchronize()
        stop, samples = threading.Event(), []

        def loop():
            while not stop.is_set():
                if hw:
                    samples.append((rd(os.path.join(hw, "power1_input")), rd(os.path.join(hw, "freq1_input"))))
                time.sleep(0.01)
        th = threading.Thread(target=loop)
        th.start()
        t0 = time.time()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        stop.set()
        th.join()
        mid = [s for s in samples[len(samples) // 4:] if s[0]]
        pw = sum(p for p, _ in mid) / max(len(mid), 1) / 1e6
        fq = sum(f for _, f in mid) / max(len(mid), 1) / 1e6
        print(f"  {name:58s} {dt * 1e3:8.2f} ms  {pw:7.1f} W  {fq:7.1f} MHz", flush=True)
        return dt

    for NC, NB in ((120, 60), (120, 30)):
        print(f"{NC} convolutions (128 -> 128 @ 128x128, batch 32) and {NB} BatchNorm-backward apply passes (256x256 x 64):", flush=True)

        def only_conv():
            for _ in range(NC):
                conv(sa)

        def only_bn():
            for _ in range(NB):
                bn(sa)

        def serial():
            k = NC // NB
            for i in range(NB):
                for _ in range(k):
                    conv(sa)
                bn(sa)

        def two():
            k = NC // NB
            for i in range(NB):                  # enqueue order interleaved so that neither queue runs dry on the host side
                for _ in range(k):
                    conv(sa)
                bn(sb)

        tc = timed("convolutions alone", only_conv)
        tb = timed("passes alone", only_bn)
        ts = timed("one stream, interleaved", serial)
        tt = timed("two streams", two)
        print(f"  => alone {1e3 * (tc + tb):.2f} ms, one stream {1e3 * ts:.2f} ms, two streams {1e3 * tt:.2f} ms "
              f"({100 * (1 - tt / ts):.1f} % less than one stream)", flush=True)


if __name__ == "__main__":
    main()
