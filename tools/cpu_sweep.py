import sys, time, os, torch
sys.path.insert(0, '.')
from ammcnet_aaai2021_amd import synthetic as S
from oracle import ammc_oracle as O
print(os.cpu_count(), os.sched_getaffinity(0).__len__())
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket'")
sd = S.make_twostream_state(n_embed=2000)
rgb_x, op_x, _, _ = S.make_clips(2, 256, 256, tag="bench")
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        O.twostream_forward(sd, rgb_x, op_x, 2)
        t0 = time.perf_counter(); O.twostream_forward(sd, rgb_x, op_x, 2); t = time.perf_counter() - t0
    print(th, 'threads', round(t, 3), 's per B=2 forward', round(2 / t, 3), 'frames/s', flush=True)
