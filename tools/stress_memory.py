"""BASELINE.json config 5: memory addressing alone, 8192 slots x 512-d, N = B*1024 feature rows."""
import sys, json, time
sys.path.insert(0, '.')
import torch
from ammcnet_aaai2021_amd import ops, synthetic as S, _lib
from ammcnet_aaai2021_amd.engine import _Packer, _ptr
dev = "cuda:0"
d, m, k = 512, 8192, 2
embed = S.hashed_normal("stress:e", (d, m), 0.9).to(dev)
lib = _lib.load()
for B in (16, 256):
    n = B * 1024
    x = (torch.randn(n, d, device=dev) * 0.8)
    s = torch.cuda.current_stream().cuda_stream
    mpad = (m + 31) // 32 * 32
    e_kblk = torch.empty((d // 8, mpad, 8), device=dev, dtype=torch.float16)
    enorm16 = torch.empty(m, device=dev)
    e_md, _ = _Packer(torch.device(dev)).codebook(embed)
    lib.ammc_pack_codebook_f16(_ptr(embed), d, m, e_kblk.data_ptr(), _ptr(enorm16), s)
    idx = torch.empty((n, k), device=dev, dtype=torch.int32)
    qk = torch.empty((n, k * d), device=dev); q1 = torch.empty((n, d), device=dev)
    part = torch.empty(lib.ammc_memory_topk_f16_blocks(n), device=dev)
    def run():
        rc = lib.ammc_memory_topk_fwd_f16(_ptr(x), e_kblk.data_ptr(), _ptr(e_md), _ptr(enorm16), n, d, m, k,
                                          idx.data_ptr(), _ptr(qk), _ptr(q1), _ptr(part), s)
        assert rc == 0, rc
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * n * d * m
    print(json.dumps({"config": "stress 8192 slots x 512-d, k=2, fp16 MFMA", "B": B, "rows": n, "ms": round(ms, 3),
                      "tflops": round(flops / ms / 1e9, 1), "frac_of_fp16_dense_peak_2500": round(flops / ms / 1e9 / 2500, 3),
                      "rows_per_s": round(n / ms * 1e3)}))
