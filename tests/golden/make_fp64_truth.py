"""The fp64 ground truth of the TIMED training batch (BASELINE.json configs[2]: 32 clips at 256x256).

What it is: the build's own oracle (`oracle/ammc_oracle.py`, pinned to the reference by `make_golden.py` /
`tests/test_oracle_golden.py`) evaluated in FLOAT64 - forward, `generator_loss`, autograd - on the clips and parameters of
`tests/golden/twostream_256_b32_train.npz` (which holds what the REFERENCE's own fp32 forward + autograd returns for them).
It answers the question the fp32 vectors cannot: of two fp32-accurate evaluations that differ by 2e-3 in a gradient norm,
which one is farther from the truth?  (`tests/test_gpu_train.py::test_batch32_gradients_are_as_close_to_fp64_as_the_reference`,
`bench.py --mode train`: `train.parity.vs_fp64`.)

Where it runs: on the GPU box's HOST cores (`--device cpu`, the default: ~60 GB of fp64 autograd state, which the
authoring container does not have; tens of minutes of CPU time) and, with `--device cuda`, the same oracle code on the
device in fp64 as a cross-check of the two evaluations (they agree to ~1e-12; recorded in the file).  Nothing of the
reference is needed: the oracle and `ammcnet_aaai2021_amd.synthetic` travel.

    python tests/golden/make_fp64_truth.py [--device cpu|cuda|both] [--out gpurun_out/twostream_256_b32_train_fp64.npz]

Stored per gradient tensor: its fp64 norm and DENSE strided samples (4096, the whole tensor when smaller; the same
positions as `gs4k.*` of the reference's fixture), plus the loss.  Also the oracle's own fp32 evaluation on this machine
at the same positions (`gs32.*`, `gn32.*`): a second fp32 witness beside the reference's.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from ammcnet_aaai2021_amd import synthetic as S  # noqa: E402
from oracle import ammc_oracle as O  # noqa: E402

DENSE = 4096


def dense_samples(g: torch.Tensor, n: int = DENSE) -> torch.Tensor:
    """the sample positions shared with make_golden.py's `gs4k.*`"""
    return g.flatten()[:: max(1, g.numel() // n)][:n].contiguous()


def oracle_step(sd, clips, dtype, device, force_idx=None, want_idx=False):
    """one G-only training step of the oracle: loss, gradients (dict name -> tensor on `device`).  `force_idx`: the
    memory lookups to take ({"rgb": [N, k], "op": [N, k]}: oracle.quantize_topk, test instrumentation); `want_idx`: also
    return the lookups this evaluation made"""
    rgb_x, op_x, rgb_t, op_t = (t.to(device=device, dtype=dtype) for t in clips)
    m = O.clone_state({k: (v.to(device=device, dtype=dtype) if v.is_floating_point() else v.to(device)) for k, v in sd.items()},
                      requires_grad=True)
    out = O.twostream_forward(m, rgb_x, op_x, 2, training=True, want_aux=want_idx, force_idx=force_idx)
    loss = O.generator_loss(out, rgb_t, op_t)
    loss.backward()
    grads = {k: v.grad.detach() for k, v in m.items() if v.requires_grad}
    if want_idx:
        return float(loss.detach()), grads, {p: out[-1][f"{p}.idx"].reshape(-1, 2) for p in ("rgb", "op")}
    return float(loss.detach()), grads


def add_reference_branch(args, cfg):
    """Two fp32-accurate evaluations of this step differ from the fp64 evaluation mostly by WHICH WAY one or two near-tie
    memory lookups fall (tools/flip_count.py: one re-routed lookup of 32768 moves the bottleneck by 5e-3 and every
    gradient downstream by ~1e-2).  Entry-by-entry comparisons therefore use the truth on the branch the evaluation
    under test took: here the reference's (its lookups are recorded in its fixture)."""
    ref = np.load(args.fixture)
    old = np.load(args.add_reference_branch)
    out = {k: old[k] for k in old.files}
    dev = "cuda" if args.device in ("cuda", "both") else "cpu"
    idx = {p: torch.as_tensor(ref[f"idx.{p}"].astype(np.int64)) for p in ("rgb", "op")}
    sd = S.make_twostream_state()
    clips = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])
    t0 = time.time()
    loss, g, own = oracle_step(sd, clips, torch.float64, dev, want_idx=True)            # unconstrained: which lookups differ?
    differ = {p: int((own[p].cpu() != idx[p]).any(dim=1).sum()) for p in idx}
    worst = max(float((dense_samples(v).cpu() - torch.as_tensor(old[f"gs64.{k}"])).norm() /
                      torch.as_tensor(old[f"gs64.{k}"]).norm().clamp_min(1e-300)) for k, v in g.items())
    del g
    loss_r, g = oracle_step(sd, clips, torch.float64, dev, force_idx=idx)
    for p_, ix in own.items():                                   # the lookups of the unconstrained fp64 evaluation
        out[f"idx64.{p_}"] = ix.cpu().numpy().astype(np.int16)
    out["loss64r"] = np.float64(loss_r)
    for k, v in g.items():
        out[f"gn64r.{k}"] = np.float64(v.norm().item())
        out[f"gs64r.{k}"] = dense_samples(v).cpu().numpy()
    meta = json.loads(str(old["meta"]))
    meta.update(reference_branch_device=dev, reference_branch_seconds=round(time.time() - t0, 1),
                reference_lookups_that_differ_from_fp64=differ, unconstrained_rerun_vs_file_max_l2rel=worst)
    out["meta"] = np.array(json.dumps(meta))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, meta, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu", choices=("cpu", "cuda", "both"))
    ap.add_argument("--fixture", default=os.path.join(HERE, "twostream_256_b32_train.npz"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "twostream_256_b32_train_fp64.npz"))
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--add-reference-branch", default="",
                    help="an existing truth file: append the truth ON THE REFERENCE'S BRANCH (`gs64r.*`, `gn64r.*`: the oracle "
                         "in fp64 with the memory lookups the reference itself made, `idx.*` of the reference fixture), "
                         "evaluated on --device, and write --out")
    args = ap.parse_args()
    cfg = json.loads(str(np.load(args.fixture)["cfg"]))
    if args.add_reference_branch:
        return add_reference_branch(args, cfg)
    if args.threads:
        torch.set_num_threads(args.threads)
    sd = S.make_twostream_state()
    clips = S.make_clips(cfg["batch"], cfg["hw"], cfg["hw"], tag=cfg["tag"])
    out = {"cfg": np.array(json.dumps(cfg)), "dense": np.int64(DENSE)}
    meta = {"threads": torch.get_num_threads(), "cpus": os.cpu_count(), "torch": torch.__version__}
    primary = None
    for dev in (("cpu", "cuda") if args.device == "both" else (args.device,)):
        t0 = time.time()
        loss, g = oracle_step(sd, clips, torch.float64, dev)
        if dev == "cuda":
            torch.cuda.synchronize()
        meta[f"seconds_fp64_{dev}"] = round(time.time() - t0, 1)
        print(f"fp64 on {dev}: {meta[f'seconds_fp64_{dev}']} s, loss {loss!r}", flush=True)
        if primary is None:
            primary = dev
            out["loss64"] = np.float64(loss)
            for k, v in g.items():
                out[f"gn64.{k}"] = np.float64(v.norm().item())
                out[f"gs64.{k}"] = dense_samples(v).cpu().numpy()
        else:       # the second evaluation: how far apart are two fp64 evaluations of the same step?
            worst = 0.0
            for k, v in g.items():
                a = torch.as_tensor(out[f"gs64.{k}"])
                worst = max(worst, float((dense_samples(v).cpu() - a).norm() / a.norm().clamp_min(1e-300)))
            meta[f"fp64_{dev}_vs_{primary}_max_l2rel"] = worst
            meta[f"fp64_{dev}_vs_{primary}_loss_rel"] = abs(loss - float(out["loss64"])) / abs(float(out["loss64"]))
            print("second fp64 evaluation vs first:", worst, flush=True)
        del g
    # the oracle's own fp32 evaluation on this machine: a second fp32 witness next to the reference's recorded one
    t0 = time.time()
    loss32, g32 = oracle_step(sd, clips, torch.float32, "cpu" if primary == "cpu" else "cuda")
    meta["seconds_fp32"] = round(time.time() - t0, 1)
    out["loss32"] = np.float64(loss32)
    for k, v in g32.items():
        out[f"gn32.{k}"] = np.float64(v.double().norm().item())
        out[f"gs32.{k}"] = dense_samples(v).cpu().numpy()
    meta["primary"] = primary
    out["meta"] = np.array(json.dumps(meta))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, meta, flush=True)


if __name__ == "__main__":
    main()
