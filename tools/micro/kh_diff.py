"""compare the KH instance with the tap-by-tap instance on one layer: where do they differ?"""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, '.')
import torch
if len(sys.argv) > 6:       # child: run one configuration, save the output
    from ammcnet_aaai2021_amd import _lib
    import tools.conv_bench_lib as cb
    B, H, W, cin, n = (int(v) for v in sys.argv[1:6])
    lib = _lib.load()
    d, keep = cb.make_desc(B, H, W, cin, n)
    _lib.check(lib.ammc_conv_gemm_s16(C.byref(d), torch.cuda.current_stream().cuda_stream), "conv")
    torch.cuda.synchronize()
    torch.save(keep[1].buf.cpu(), sys.argv[6])
    sys.exit(0)
args = sys.argv[1:6]
for kh, f in ((0, "/tmp/o0.pt"), (1, "/tmp/o1.pt")):
    subprocess.run([sys.executable, __file__] + args + [f], env=dict(os.environ, AMMC_TAP_KH=str(kh), AMMC_S16_MF="0"), check=True)
a, b = torch.load("/tmp/o0.pt"), torch.load("/tmp/o1.pt")
def dec(t):
    hv = t.view(torch.float16).view(*t.shape[:-1], t.shape[-1] // 8, 2, 8).float()
    return (hv[..., 0, :] + hv[..., 1, :] / 2048.0).reshape(*t.shape[:-1], -1)
fa, fb = dec(a), dec(b)
err = (fa - fb).abs()
print("decoded: max |a| %.4g  max |a-b| %.4g  rel %.3g   mean|a-b| %.3g" % (fa.abs().max(), err.max(), err.max() / fa.abs().max(), err.mean()))
big = err > 1e-3 * fa.abs().max()
print("elements off by > 1e-3 of max:", int(big.sum()), "of", big.numel())
if big.any():
    print("  channels:", big.sum((0, 1, 2)).nonzero().flatten().tolist()[:64])
    print("  rows:", big[0].sum((1, 2)).nonzero().flatten().tolist()[:20], " cols:", big[0].sum((0, 2)).nonzero().flatten().tolist()[:40])
    i0 = big.nonzero()[0].tolist(); print("  e.g.", i0, float(fa[tuple(i0)]), float(fb[tuple(i0)]))
ai, bi = a.view(torch.int32), b.view(torch.int32)
bad = (ai != bi)
print("shape", tuple(a.shape), "differing words", int(bad.sum()), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()
    print("first", idx[:5].tolist())
    print("per batch", bad.sum((1, 2, 3)).tolist()[:8])
    print("rows (y) with errors (first image)", bad[0].sum((1, 2)).nonzero().flatten().tolist()[:40])
    print("cols (x) with errors (first image)", bad[0].sum((0, 2)).nonzero().flatten().tolist()[:40])
    ch = bad[0].sum((0, 1))
    print("float index in the channel dim with errors", ch.nonzero().flatten().tolist())
