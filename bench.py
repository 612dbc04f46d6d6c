"""Benchmark of the hot path: frames/s of `twostream.forward` on synthetic 256x256 clips.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Ped2-shaped dual-stream network with a 2000-slot memory
(embed_dim 64, k 2), inference, batch 16 per GPU, inputs resident in HBM.  One "step" = one
forward over one batch.  N > 1 = N independent replicas on disjoint clips (the reference has
no multi-GPU semantics; inference shards by whole batches and needs no collective), so
`scaling` is "weak" and `value` is the sum over ranks.

Besides the contract fields the JSON line carries
  roofline      algorithmic FLOPs of the dominant kernel (S16: conv_tap_s16<4, 1, 2, 4, 1>; --precision fp32:
                conv_gemm_f32<128x128>) per launch divided by its average launch duration (HIP events on the
                launch stream), against the dense fp16 (2500) / fp32 (157.3 TFLOP/s) MFMA peak of MI355X
  cpu_baseline  the CPU oracle (oracle/ammc_oracle.py, "port") timed on this host on a
                bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=16, help="clips per GPU per step")
    p.add_argument("--n-embed", type=int, default=2000)
    p.add_argument("--size", type=int, default=256)
    p.add_argument("--precision", choices=("fp32", "s16"), default=os.environ.get("AMMC_PRECISION", "s16"),
                   help="fp32 = exact fp32 MFMA; s16 = split-fp16 MFMA with fp32 accumulation (fp32-equivalent)")
    p.add_argument("--mode", choices=("infer", "train"), default="infer",
                   help="infer = the headline metric (BASELINE.json configs[1]); train = configs[2]/[3]: fwd+bwd+Adam, "
                        "batch 32 per GPU, data parallel with a bucketed RCCL gradient all-reduce when --gpus > 1")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-batch", type=int, default=4)
    p.add_argument("--cpu-iters", type=int, default=10)
    return p.parse_args()


def cpu_baseline(args):
    """the CPU restatement of the same forward on a bounded sample (kind "port")"""
    from ammcnet_aaai2021_amd import synthetic as S
    from oracle import ammc_oracle as O
    sd = S.make_twostream_state(n_embed=args.n_embed)
    b = args.cpu_sample_batch
    rgb_x, op_x, _, _ = S.make_clips(b, args.size, args.size, tag="bench")
    ncpu = os.cpu_count() or 1
    # ATen's CPU convolutions stop scaling (and then collapse) well before 256 threads on the
    # GPU box's 2x64-core host: take the best of a few thread counts, one forward each
    best = None
    with torch.no_grad():
        for th in sorted({min(t, ncpu) for t in (8, 16, 32)}):
            torch.set_num_threads(th)
            O.twostream_forward(sd, rgb_x, op_x, 2)                 # warm-up at this thread count
            t0 = time.perf_counter()
            O.twostream_forward(sd, rgb_x, op_x, 2)
            dt = time.perf_counter() - t0
            if best is None or dt < best[1]:
                best = (th, dt)
        threads = best[0]
        torch.set_num_threads(threads)
        times = []
        for _ in range(args.cpu_iters):
            t0 = time.perf_counter()
            O.twostream_forward(sd, rgb_x, op_x, 2)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(b / med, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{args.cpu_iters} timed forwards (median) of batch {b} at {args.size}x{args.size}, "
                      f"n_embed {args.n_embed}, torch CPU fp32, best of 8/16/32 threads = {threads} "
                      f"(host has {ncpu} logical CPUs)"}


def train_mode(args, rank, world, dev, dist):
    """BASELINE.json configs[2] / configs[3]: one optimisation step of the shipped network (256 slots) per "step",
    batch 32 per GPU (weak scaling), gradients averaged over RCCL inside backward (parallel.BucketedGradReducer)."""
    import ammcnet_aaai2021_amd as A
    from ammcnet_aaai2021_amd import harness, parallel, synthetic as S
    from oracle.ammc_oracle import fwd_flops_per_clip
    batch = 32 if args.batch == 16 else args.batch
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(S.make_twostream_state())
    net = net.to(dev).train()
    if world > 1:
        parallel.broadcast_state(net)
        parallel.attach_reducer(net, parallel.BucketedGradReducer())
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    rgb_x, op_x, rgb_t, op_t = (t.to(dev) for t in S.make_clips(batch, args.size, args.size, tag=f"trainbench{rank}"))
    rgb = torch.cat([rgb_x.view(batch, 4, 3, args.size, args.size), rgb_t[:, None]], 1)
    op = torch.cat([op_x.view(batch, 3, 2, args.size, args.size), op_t[:, None]], 1)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        loss = harness.train_step(net, opt, rgb, op)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = harness.train_step(net, opt, rgb, op)
    barrier()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(loss))
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        value = batch * args.steps * world / elapsed
        flops = 3.0 * fwd_flops_per_clip(args.size, args.size)
        from ammcnet_aaai2021_amd import train as T
        print(json.dumps({
            "metric": "clips/sec, 256x256x4 dual-stream clips (twostream forward + backward + Adam, training)",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if T.TRAIN_PRECISION == "fp32" else "f32-equivalent: (hi,lo) f16 split MFMA for the 3x3 convolutions, f32 elsewhere",
            "data": "synthetic",
            "config": {"workload": "Ped2 dual-stream + 256-slot memory, batch 32 per GPU, fwd+bwd+Adam "
                                   "(BASELINE.json configs[2]; configs[3] with --gpus 8)",
                       "batch_per_gpu": batch, "frame": f"{args.size}x{args.size}",
                       "parallelism": f"dp{world} (bucketed RCCL all-reduce of 100 MB of fp32 gradients)"},
            "whole_path_tflops": round(value * flops / 1e12 / world, 2), "roofline": None, "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # AMMC_BENCH_SHARE_GPU=1 (tests on a one-GPU box): every rank on cuda:0, gloo instead of RCCL (which needs one
    # device per rank); everything else is the code the multi-GPU runs execute
    share = os.environ.get("AMMC_BENCH_SHARE_GPU", "0") != "0"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import ammcnet_aaai2021_amd as A
    from ammcnet_aaai2021_amd import synthetic as S

    if args.mode == "train":
        return train_mode(args, rank, world, dev, dist)

    sd = S.make_twostream_state(n_embed=args.n_embed)
    net = A.get_twostream((12, 6), (3, 2), 64, args.n_embed, 2)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    net.precision = args.precision
    # each rank works on its own clips (weak scaling: per-GPU work fixed)
    rgb_x, op_x, _, _ = S.make_clips(args.batch, args.size, args.size, tag=f"bench{rank}")
    rgb_x, op_x = rgb_x.to(dev), op_x.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = net(rgb_x, op_x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(rgb_x, op_x)
        barrier()
        elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(out[0]).all())
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel, measured live with HIP events --------------
    roof = None
    per_kernel = {}
    if rank == 0:
        eng = net._engine
        agg = {}
        reps = 3
        eng._timed = True
        for _ in range(reps):
            with torch.no_grad():
                net(rgb_x, op_x)
            for meta, ms in eng.timings:
                a = agg.setdefault(meta["kernel"] or meta["name"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                a["ms"] += ms
                a["flops"] += meta["flops"]
                a["bytes"] += meta["bytes"]
                a["launches"] += 1
        eng._timed = False
        total_ms = sum(a["ms"] for a in agg.values()) / reps
        for kname, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            per_kernel[kname] = dict(launches_per_step=a["launches"] // reps, avg_us=round(1e3 * a["ms"] / a["launches"], 2),
                                     share=round(a["ms"] / reps / total_ms, 4),
                                     tflops=round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 2) if a["flops"] else None,
                                     gbs=round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1) if a["bytes"] else None)
        dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
        a = dom[1]
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch cannot be read live: they come from the committed PMC passes
        # (tools/profile_bench.sh -> profiles/rNN_pmc_traffic.json) of this same command
        traffic = None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.precision}_pmc_traffic.json")))[-1:]:
            with open(path) as fp:
                traffic = json.load(fp)["kernels"].get(dom[0], {}).get("traffic_bytes_per_launch")
        s16 = args.precision == "s16"
        # S16: `achieved` stays ALGORITHMIC (one multiply-add per filter tap); the kernel issues 3 fp16 MFMAs
        # per algorithmic product, so frac <= 1/3 by construction against the dense fp16 peak
        peak = PEAK_F16_MFMA_TFLOPS if s16 else PEAK_F32_MFMA_TFLOPS
        roof = {"kernel": dom[0], "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "mfma_issue_frac": round((3.0 if s16 else 1.0) * achieved / peak, 4),
                "flops_per_launch": a["flops"] / a["launches"], "avg_launch_us": round(1e3 * a["ms"] / a["launches"], 2),
                "launches_per_step": a["launches"] // reps}

    if rank == 0:
        from oracle.ammc_oracle import fwd_flops_per_clip
        frames = args.batch * args.steps * world
        value = frames / elapsed
        flops_clip = fwd_flops_per_clip(args.size, args.size, n_embed=args.n_embed)
        line = {
            "metric": "frames/sec, 256x256x4 dual-stream clips (twostream forward, inference)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32-equivalent: (hi,lo) f16 split, 3x v_mfma_f32_32x32x16_f16, f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "Ped2 full dual-stream + 2000-slot memory module, batch=16, inference "
                                   "(BASELINE.json configs[1])",
                       "batch_per_gpu": args.batch, "frame": f"{args.size}x{args.size}", "n_embed": args.n_embed,
                       "embed_dim": 64, "k": 2, "parallelism": f"replicas x{world} (no collective)",
                       "gflop_per_frame": round(flops_clip / 1e9, 2)},
            "whole_path_tflops": round(value * flops_clip / 1e12 / world, 2),
            "whole_path_frac_of_f32_mfma_peak": round(value * flops_clip / 1e12 / world / PEAK_F32_MFMA_TFLOPS, 4),
            "precision": args.precision,
            "roofline": roof,
            "kernels": per_kernel,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
