"""Where a tile of the halo-patch kernel spends its time (diagnostic build: AMMC_HIPCC_FLAGS=-DAMMC_TAP_STAMP
python -m ammcnet_aaai2021_amd.build --variant stamp; AMMC_LIB=.../libammc_hip_stamp.so python tools/micro/tap_stamps.py B H W CIN N)
slots: 0 tile start, 1 prologue landed, 2+2cc end of block cc (1+2cc: its patch reload landed), 14 stores issued, 15 stores acknowledged"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import _lib
import tools.conv_bench_lib as cb

B, H, W, cin, n = (int(v) for v in sys.argv[1:6])
lib = _lib.load()
d, keep = cb.make_desc(B, H, W, cin, n)
tiles = B * (H // 8) * (W // 32) * max(1, n // 128)
st = torch.zeros(tiles * 16, dtype=torch.int64, device="cuda:0")
C.CDLL(_lib.LIB_PATH).ammc_debug_set_tap_stamps(C.c_void_p(st.data_ptr()))
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv")
torch.cuda.synchronize()
st.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(lib.ammc_conv_gemm_s16(C.byref(d), s), "conv")
e1.record()
torch.cuda.synchronize()
t = st.view(tiles, 16).cpu().double() / 100.0          # us
t0 = t[:, 0].min()
ncc = min(cin // 32, 6)
print(f"B={B} {H}x{W} {cin}->{n}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {tiles} tiles")
start = t[:, 0] - t0
rounds = torch.unique((start / (start.max() / max(1, tiles // 512) + 1e-9)).floor())
print(f"tile starts: min {start.min():.1f} max {start.max():.1f};  ends (stores issued) max {(t[:, 14] - t0).max():.1f}, acked max {(t[:, 15] - t0).max():.1f}")
def col(i): return t[:, i]
seg = [("prologue (patch + first slices landed)", col(1) - col(0))]
prev = col(1)
for cc in range(ncc):
    if cc > 0 and (col(1 + 2 * cc) > 0).all():
        seg.append((f"block {cc}: patch reload", col(1 + 2 * cc) - prev))
        prev = col(1 + 2 * cc)
    seg.append((f"block {cc}: nine taps", col(2 + 2 * cc) - prev))
    prev = col(2 + 2 * cc)
if (col(10) > 0).all():
    seg.append(("epilogue: until the first channel group", col(10) - prev))
    for j in range(1, 4):
        if (col(10 + j) > 0).all():
            seg.append((f"epilogue: 32-filter group {j - 1}", col(10 + j) - col(9 + j)))
seg.append(("epilogue until the last store is issued", col(14) - prev))
seg.append(("stores acknowledged (vmcnt 0)", col(15) - col(14)))
seg.append(("whole tile", col(15) - col(0)))
for name, v in seg:
    print(f"  {name:42s} mean {v.mean():7.2f}  median {v.median():7.2f}  p10 {v.quantile(0.1):7.2f}  p90 {v.quantile(0.9):7.2f} us")
import os
if os.environ.get("STAMP_RAW"):
    # raw timelines of the first-round CU-mates (b, b + 256) and whoever follows them
    for b in (0, 256, 8, 264, 100, 356):
        r = (t[b] - t0).tolist()
        print(f"tile {b:4d}: " + " ".join(f"{v:7.1f}" if v > -1e6 else "      -" for v in r))
    late = torch.argsort(t[:, 0])[512:520].tolist()
    for b in late:
        r = (t[b] - t0).tolist()
        print(f"tile {b:4d}: " + " ".join(f"{v:7.1f}" if v > -1e6 else "      -" for v in r))
