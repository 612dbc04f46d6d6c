"""End-to-end evaluation entry, the counterpart of `python -m Code.main.run_test` (reference
Code/main/run_test.py:10-23 -> run_helper/test_helper.py:519-570): build the model, load a checkpoint,
score every sub-video with the reference's loop semantics, fuse the scores, report fps and AUC.

The reference reads jpg/.flo folders and ground-truth .mat files that do not exist here, so
the data source is either tensors saved with torch.save ({"videos": [(rgb[T,3,H,W], flow[T-1,2,H,W]), ...],
"gt": [labels[T], ...]}) or `--synthetic`, a deterministic stand-in with the reference's value ranges.

    python -m ammcnet_aaai2021_amd.run_test --synthetic --dataset_name ped2 [--ckpt model.pth] [--precision s16]
    python -m ammcnet_aaai2021_amd.run_test --synthetic --raw          # through the input pipeline (uint8 frames + flows)
    python -m ammcnet_aaai2021_amd.run_test --rgb_root DIR --op_root DIR   # the reference's folder layout (jpg/png/.npy + .flo)
    torchrun --nproc-per-node 8 -m ammcnet_aaai2021_amd.run_test --synthetic      # whole batches sharded over GPUs
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np
import torch

from . import harness, parallel, pipeline, synthetic
from .unet import get_twostream


def synthetic_dataset(n_videos: int, frames: int, size: int):
    vids, gts = [], []
    for i in range(n_videos):
        t = frames + 7 * (i % 3)
        rgb = synthetic.hashed_uniform(f"rt:rgb{i}", (t, 3, size, size))
        u = synthetic.hashed_normal(f"rt:op{i}", (t - 1, 1, size, size), 2.0) / 256.0
        vids.append((rgb, torch.cat([u, u / 256.0], 1)))
        gts.append((synthetic.hashed_uniform(f"rt:gt{i}", (t,), 0, 1) > 0.8).numpy().astype(np.int8))
    return vids, gts


def synthetic_raw_sources(n_videos: int, frames: int, h: int = 240, w: int = 360):
    """decoded-frame stand-ins at a camera resolution (ped2 is 240x360): uint8 RGB frames and .flo-like flows"""
    def make(i):
        t = frames + 7 * (i % 3)
        rgb = ((synthetic.hashed_uniform(f"raw:rgb{i}", (t, h, w, 3)) + 1) * 127.5).round().clamp(0, 255).to(torch.uint8)
        flow = synthetic.hashed_normal(f"raw:op{i}", (t - 1, h, w, 2), 2.0)
        return rgb.numpy(), flow.numpy()
    gts = [(synthetic.hashed_uniform(f"rt:gt{i}", (frames + 7 * (i % 3),), 0, 1) > 0.8).numpy().astype(np.int8)
           for i in range(n_videos)]
    return [(lambda i=i: make(i)) for i in range(n_videos)], gts


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--dataset_name", default="ped2", choices=sorted(harness.LAM_MAP))
    p.add_argument("--ckpt", default=None, help="state_dict of the reference's twostream generator (.pth)")
    p.add_argument("--data", default=None, help="torch.save'd dict with 'videos' and optional 'gt'")
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--raw", action="store_true", help="with --synthetic: uint8 frames + flows through pipeline.SubVideoStager")
    p.add_argument("--rgb_root", default=None, help="folder of sub-video folders of frames (jpg/png/.npy)")
    p.add_argument("--op_root", default=None, help="folder of sub-video folders of flows (.flo/.npy)")
    p.add_argument("--videos", type=int, default=4)
    p.add_argument("--frames", type=int, default=60)
    p.add_argument("--size", type=int, default=256)
    p.add_argument("--precision", choices=("fp32", "s16"), default="s16")
    p.add_argument("--embed_dim", type=int, default=64)
    p.add_argument("--n_embed", type=int, default=256)
    p.add_argument("--k", type=int, default=2)
    a = p.parse_args(argv)

    rank, world, dev = parallel.init_distributed()
    if dev.type != "cuda":
        raise SystemExit("run_test needs a GPU: the HIP path has no CPU fallback")
    model = get_twostream((12, 6), (3, 2), a.embed_dim, a.n_embed, a.k)
    if a.ckpt:
        model.load_state_dict(torch.load(a.ckpt, map_location="cpu"), strict=True)      # test_helper.py:558
    else:
        model.load_state_dict(synthetic.make_twostream_state(embed_dim=a.embed_dim, n_embed=a.n_embed, k=a.k))
    model = model.to(dev).eval()
    model.precision = a.precision
    staged = None
    if a.data:
        blob = torch.load(a.data, map_location="cpu")
        videos, gt = blob["videos"], blob.get("gt")
    elif a.rgb_root and a.op_root:
        sources, gt = pipeline.list_subvideos(a.rgb_root, a.op_root), None
    elif a.synthetic and a.raw:
        sources, gt = synthetic_raw_sources(a.videos, a.frames)
    elif a.synthetic:
        videos, gt = synthetic_dataset(a.videos, a.frames, a.size)
    else:
        raise SystemExit("give --data, --rgb_root/--op_root or --synthetic")
    if a.rgb_root and a.op_root or (a.synthetic and a.raw):
        # raw inputs STREAM through the loop (round 5): sub-video v + 1 / v + 2 are read, uploaded and resized / normalised
        # on a side stream while v is scored (`pipeline.SubVideoStager`, `harness.evaluate_stream`); one sub-video's
        # frames are resident at a time, every batch of clips is a window into them.  Ranks take every world-th sub-video.
        def staged_loop(srcs):
            st = pipeline.SubVideoStager(srcs, dev, size=(a.size, a.size), shard=(rank, world), ahead=2)
            lens = []

            def feed():
                for rgb, op in st:
                    lens.append(rgb.shape[0])
                    yield rgb, op
            rec = harness.evaluate_stream(model, feed(), a.dataset_name)
            return rec, st, lens
        staged_loop(sources[:max(world, 1)])                                     # warm-up: plans, packs, pinned pools
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.time()
        rec, st, lens = staged_loop(sources)
        torch.cuda.synchronize()
        if world > 1:                     # the clock stops when the SLOWEST rank is done: fps = all ranks' frames / that time
            torch.distributed.barrier()
        used = time.time() - t0
        staged = {"host_read_s": round(st.host_seconds, 3), "uploaded_MB": round(st.bytes_uploaded / 2**20, 1),
                  "staging": "streamed: overlapped with the scoring loop (total_time_s covers it)"}
        if world > 1:                                                            # sub-video v lives on rank v % world
            parts = [None] * world
            torch.distributed.all_gather_object(parts, (rec, lens))
            keys = [k for k in rec if k != "dataset"]
            merged = {"dataset": rec["dataset"], **{k: [] for k in keys}}
            lens_all = []
            for v in range(len(sources)):
                r, ln = parts[v % world]
                for k in keys:
                    merged[k].append(r[k][v // world])
                lens_all.append(ln[v // world])
            rec, lens = merged, lens_all
        n_frames = lens
    else:
        harness.evaluate_dataset(model, videos[:1], a.dataset_name, device=dev)          # warm-up: plans, packs
        torch.cuda.synchronize()
        t0 = time.time()
        rec = harness.evaluate_dataset(model, videos, a.dataset_name, device=dev, rank=rank, world=world)
        torch.cuda.synchronize()
        used = time.time() - t0
        n_frames = [v[0].shape[0] for v in videos]
    if rank == 0:
        n_pred = sum(t - harness.RGB_LEN_CLIP + 1 for t in n_frames)
        out = {"dataset": a.dataset_name, "videos": len(n_frames), "predicted_frames": n_pred, "gpus": world,
               "total_time_s": round(used, 3), "fps": round(n_pred / used, 2), "precision": a.precision,
               "s16_fallbacks": getattr(model, "s16_fallbacks", 0)}                      # test_helper.py:485-486
        if staged:
            out.update(staged)
        if gt is not None:
            out["auc"] = harness.fuse_scores_auc(rec, gt)["auc"]
            out["auc_note"] = "synthetic labels" if a.synthetic else "labels from --data"
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
