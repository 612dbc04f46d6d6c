"""memory_topk_s16 at 16384 rows x 2000 slots: 32-row against 64-row workgroups (option "memory_rt")"""
import sys
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import _lib
from ammcnet_aaai2021_amd.engine import _ptr
lib = _lib.load()
dev = "cuda:0"
d, k = 64, 2
s = torch.cuda.current_stream().cuda_stream
for n in (16384, 32768):
    x = torch.randn(n, d, device=dev) * 0.8
    for m in (256, 2000):
        e = torch.randn(d, m, device=dev) * 0.9
        mpad = (m + 31) // 32 * 32
        e_md = torch.empty(m, d, device=dev)
        enorm = torch.empty(m, device=dev)
        _lib.check(lib.ammc_pack_codebook_f32(_ptr(e), d, m, _ptr(e_md), _ptr(enorm), s), "pack")
        e16 = torch.empty((d // 8, mpad, 16), device=dev, dtype=torch.float16)
        _lib.check(lib.ammc_pack_codebook_s16(_ptr(e), d, m, e16.data_ptr(), s), "pack16")
        idx = torch.empty((n, k), device=dev, dtype=torch.int32)
        qk = torch.empty((n, k * d), device=dev)
        q1 = torch.empty((n, d), device=dev)
        part = torch.empty(lib.ammc_memory_topk_blocks(n), device=dev)
        for rt in (1, 2):
            lib.ammc_set_option(b"memory_rt", rt)
            def run():
                return lib.ammc_memory_topk_fwd_s16(_ptr(x), e16.data_ptr(), _ptr(e_md), _ptr(enorm), n, d, m, k, idx.data_ptr(), _ptr(qk), _ptr(q1), _ptr(part), s)
            for _ in range(5):
                _lib.check(run(), "run")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            print(f"n={n} m={m:5d} rows/workgroup={32 * rt}: {e0.elapsed_time(e1) * 20:.1f} us per launch", flush=True)
lib.ammc_set_option(b"memory_rt", 0)
