"""Host-side counterparts of the two callers of the path (the reference's harness files do not
travel and are not importable without cv2/tensorboardX).

* `evaluate_subvideo` / `evaluate_dataset`: the scoring loop of
  `run_helper/test_helper.py:408-488` - sliding clips in order, batches of 16 (last one
  short), per-sample PSNR (`utils/utils.py:130-148`), ONE commit value per batch written to
  every frame of the batch, first `len_clip-1` frames back-filled - and the record dict the
  reference pickles (`:479-484`).  Multi-GPU: whole batches are sharded across ranks
  (`parallel.shard_batches`), never split.
* `generator_loss` / `train_step`: the G-only objective of SURVEY.md 3.2 (the part of
  `Twostream_vq_Loss`, `models/losses/loss_zoo.py:323-336`, that exercises the path's backward):
  lam_lp * L2norm(rgb) + lam_lp_op * L2norm(op) + lam_latent * (rgb_diff + op_diff).

The model is any callable with the reference's forward signature; with the HIP modules the
loop makes one device->host copy per BATCH (the reference syncs once per frame, `:452-453`).
"""
from __future__ import annotations

import contextlib
import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import parallel

EVAL_BATCH = 16            # hard-coded in the reference (test_helper.py:414-417)
RGB_LEN_CLIP, OP_LEN_CLIP = 5, 4


def psnr_per_sample(gen: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """[B] PSNR on the [0,1] range (utils/utils.py:141-148 applied to each sample)"""
    n = gen.shape[1] * gen.shape[2] * gen.shape[3]
    sq = ((gt + 1.0) / 2.0 - (gen + 1.0) / 2.0) ** 2
    return 10.0 * torch.log10(1.0 / ((1.0 / n) * sq.sum(dim=[1, 2, 3])))


def subvideo_batches(n_frames: int, batch: int = EVAL_BATCH, rgb_len_clip: int = RGB_LEN_CLIP):
    """[(first clip, last clip + 1)] of one sub-video, in order, last batch short"""
    n_clip = n_frames - rgb_len_clip + 1
    return [(s, min(s + batch, n_clip)) for s in range(0, max(n_clip, 0), batch)]


def clip_windows(frames: torch.Tensor, s: int, e: int, n_in: int) -> torch.Tensor:
    """clips [s, e) of a contiguous [T, c, H, W] sub-video as ONE overlapping view [e - s, n_in * c, H, W]: clip i is
    `frames[i:i + n_in].view(n_in * c, H, W)` (test_helper.py:433-438), so consecutive clips are one frame apart"""
    _, c, h, w = frames.shape
    return frames.as_strided((e - s, n_in * c, h, w), (c * h * w, h * w, w, 1), frames.storage_offset() + s * c * h * w)


def score_batch_device(model: Callable, rgb_frames: torch.Tensor, op_frames: torch.Tensor, s: int, e: int,
                       device=None, exact: bool = False) -> torch.Tensor:
    """run clips [s, e) of a sub-video as ONE batch; returns [2 b + 3] on the model's device: per-clip rgb PSNR, per-clip
    flow PSNR, the batch's two commit values and the S16 range flag of the batch (0 / 1; always 0 for models without
    one) - nothing is copied to the host, so batches can be queued back to back.  `exact`: run the HIP model's
    exact-fp32 kernels (the re-run of a flagged batch)."""
    b = e - s
    resident = device is None or (rgb_frames.device == torch.device(device) and op_frames.device == torch.device(device))
    if resident and rgb_frames.is_contiguous() and op_frames.is_contiguous():
        # the sub-video is where the model is: a batch of clips is an OVERLAPPING view of it (clip i = frames [i, i + 4),
        # batch stride = one frame) and the targets are a plain slice - nothing is gathered, the HIP model's first-layer
        # kernel reads the windows in place (ammc_conv_first_s16_bs); other callables make them contiguous themselves
        rgb_in = clip_windows(rgb_frames, s, e, RGB_LEN_CLIP - 1)
        op_in = clip_windows(op_frames, s, e, OP_LEN_CLIP - 1)
        rgb_t = rgb_frames[s + RGB_LEN_CLIP - 1:e + RGB_LEN_CLIP - 1]
        op_t = op_frames[s + OP_LEN_CLIP - 1:e + OP_LEN_CLIP - 1]
    else:
        rgb = torch.stack([rgb_frames[i:i + RGB_LEN_CLIP] for i in range(s, e)])
        op = torch.stack([op_frames[i:i + OP_LEN_CLIP] for i in range(s, e)])
        if device is not None:
            rgb, op = rgb.to(device, non_blocking=True), op.to(device, non_blocking=True)
        rgb_in = rgb[:, :-1].reshape(b, -1, *rgb.shape[-2:])
        op_in = op[:, :-1].reshape(b, -1, *op.shape[-2:])
        rgb_t, op_t = rgb[:, -1], op[:, -1]
    with torch.no_grad():
        flag = None
        if hasattr(model, "forward_scored") and rgb_in.is_cuda:
            # the HIP model accumulates the squared errors inside the `outc` kernel
            (_, _, (rgb_diff, op_diff), _), rgb_psnr, op_psnr = model.forward_scored(
                rgb_in, op_in, rgb_t, op_t, defer_guard=True, exact=exact)
            flag = getattr(model, "last_overflow", None)
        else:
            rgb_out, op_out, (rgb_diff, op_diff), _ = model(rgb_in, op_in)
            rgb_psnr, op_psnr = psnr_per_sample(rgb_out, rgb_t), psnr_per_sample(op_out, op_t)
        if flag is None:
            flag = torch.zeros(1, device=rgb_psnr.device, dtype=rgb_psnr.dtype)
        return torch.cat([rgb_psnr, op_psnr, rgb_diff.reshape(1), op_diff.reshape(1), flag.reshape(1).to(rgb_psnr.dtype)])


def _unpack_scores(stats: np.ndarray, b: int) -> Dict[str, np.ndarray]:
    return {"rgb_psnr": stats[:b], "op_psnr": stats[b:2 * b], "rgb_comm": stats[2 * b], "op_comm": stats[2 * b + 1],
            "overflow": bool(stats[2 * b + 2] != 0)}


def score_batch(model: Callable, rgb_frames: torch.Tensor, op_frames: torch.Tensor, s: int, e: int,
                device=None) -> Dict[str, np.ndarray]:
    """`score_batch_device` + the copy to the host (one sync); a batch whose S16 range flag is set is re-run on the
    exact-fp32 kernels"""
    sc = _unpack_scores(score_batch_device(model, rgb_frames, op_frames, s, e, device).cpu().numpy(), e - s)
    if sc["overflow"]:
        sc = _unpack_scores(score_batch_device(model, rgb_frames, op_frames, s, e, device, exact=True).cpu().numpy(), e - s)
        _count_fallback(model)
    return sc


def _count_fallback(model) -> None:
    if isinstance(model, torch.nn.Module):
        object.__setattr__(model, "s16_fallbacks", getattr(model, "s16_fallbacks", 0) + 1)


def assemble_records(n_frames: int, batches, scores: Sequence[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    """per-frame arrays of one sub-video from its batch scores (test_helper.py:445-473)"""
    rec = {k: np.empty((n_frames,), dtype=np.float32) for k in ("rgb_psnr", "rgb_comm", "op_psnr", "op_comm")}
    for (s, e), sc in zip(batches, scores):
        for i in range(s, e):
            rec["rgb_psnr"][i + RGB_LEN_CLIP - 1] = sc["rgb_psnr"][i - s]
            rec["rgb_comm"][i + RGB_LEN_CLIP - 1] = sc["rgb_comm"]
            rec["op_psnr"][i + OP_LEN_CLIP - 1] = sc["op_psnr"][i - s]
            rec["op_comm"][i + OP_LEN_CLIP - 1] = sc["op_comm"]
    for key, lc in (("rgb_psnr", RGB_LEN_CLIP), ("rgb_comm", RGB_LEN_CLIP), ("op_psnr", OP_LEN_CLIP),
                    ("op_comm", OP_LEN_CLIP)):
        rec[key][:lc - 1] = rec[key][lc - 1]
    rec["op_psnr"][n_frames - 1] = rec["op_psnr"][n_frames - 2]
    rec["op_comm"][n_frames - 1] = rec["op_comm"][n_frames - 2]
    return rec


def evaluate_subvideo(model: Callable, rgb_frames: torch.Tensor, op_frames: torch.Tensor, device=None):
    """rgb_frames [T,3,H,W], op_frames [T-1,2,H,W] -> per-frame record arrays"""
    t = rgb_frames.shape[0]
    if op_frames.shape[0] != t - 1:
        raise ValueError("a sub-video of T frames has T-1 flows")
    batches = subvideo_batches(t)
    scores = [score_batch(model, rgb_frames, op_frames, s, e, device) for s, e in batches]
    return assemble_records(t, batches, scores)


def evaluate_dataset(model: Callable, videos: Sequence, dataset_name: str = "synthetic", device=None,
                     rank: int = 0, world: int = 1) -> Optional[dict]:
    """videos: sequence of (rgb_frames, op_frames) in sorted sub-video order.  Whole batches are
    sharded over `world` ranks; every rank returns the full record dict (test_helper.py:479-484)."""
    plan = [(v, s, e) for v, (rgb, _) in enumerate(videos) for s, e in subvideo_batches(rgb.shape[0])]
    mine = parallel.shard_batches(len(plan), rank, world)
    pending = []
    resident = (None, None)                 # one sub-video at a time lives on the device (<= ~150 MB for 180 frames)
    for i in mine:
        v, s, e = plan[i]
        if device is not None and resident[0] != v:
            resident = (v, (videos[v][0].to(device, non_blocking=True), videos[v][1].to(device, non_blocking=True)))
        rgb_v, op_v = resident[1] if device is not None else videos[v]
        pending.append((i, e - s, score_batch_device(model, rgb_v, op_v, s, e, None)))      # queued, not awaited
    local = {}
    if pending:                              # ONE device-to-host copy for all batches of this rank
        flat = torch.cat([t for _, _, t in pending]).cpu().numpy()
        off = 0
        for i, b, t in pending:
            local[i] = _unpack_scores(flat[off:off + t.numel()], b)
            off += t.numel()
        # S16 range guard (zero syncs in the loop above): the per-batch flags came back with the scores; batches
        # whose activations left the half range are re-run on the exact-fp32 kernels, the rest stand
        for i in [i for i in local if local[i]["overflow"]]:
            v, s, e = plan[i]
            rgb_v, op_v = videos[v]
            local[i] = _unpack_scores(score_batch_device(model, rgb_v, op_v, s, e, device, exact=True).cpu().numpy(), e - s)
            _count_fallback(model)
    allsc = parallel.gather_records(local, world)
    out = {"dataset": dataset_name, "rgb_img_pred_records": [], "rgb_fea_comm_records": [],
           "op_img_pred_records": [], "op_fea_comm_records": []}
    for v, (rgb, _) in enumerate(videos):
        idx = [i for i, p in enumerate(plan) if p[0] == v]
        rec = assemble_records(rgb.shape[0], [plan[i][1:] for i in idx], [allsc[i] for i in idx])
        out["rgb_img_pred_records"].append(rec["rgb_psnr"])
        out["rgb_fea_comm_records"].append(rec["rgb_comm"])
        out["op_img_pred_records"].append(rec["op_psnr"])
        out["op_fea_comm_records"].append(rec["op_comm"])
    return out


def evaluate_stream(model: Callable, subvideos, dataset_name: str = "synthetic", stats: Optional[dict] = None) -> dict:
    """`evaluate_dataset` for sub-videos that ARRIVE device-resident one after the other (`pipeline.SubVideoStager`: the
    upload and the resize / normalise kernels of sub-video v + 1 run on a side stream while v is scored) - the whole
    loop of run_helper/test_helper.py:408-488 without a host wait inside it:

      * every batch of a sub-video is queued back to back (`score_batch_device`: clips are overlapping windows of the
        resident tensors, PSNR comes out of the output layer's epilogue);
      * the scores and S16 range flags of a sub-video go to pinned host memory by ONE asynchronous copy;
      * they are read one sub-video LATE - the device is then busy with the next one - and batches whose flag is set are
        re-run on the exact-fp32 kernels while their sub-video is still resident.

    `stats` (optional dict) receives `score_copies`, `rerun_batches`."""
    out = {"dataset": dataset_name, "rgb_img_pred_records": [], "rgb_fea_comm_records": [],
           "op_img_pred_records": [], "op_fea_comm_records": []}
    copies = reruns = 0

    def finish(item):
        nonlocal reruns
        rgb, op, batches, host, ev = item
        ev.synchronize()
        flat, off, scores = host.numpy(), 0, []
        for s, e in batches:
            n = 2 * (e - s) + 3
            sc = _unpack_scores(flat[off:off + n].copy(), e - s)
            off += n
            if sc["overflow"]:
                sc = _unpack_scores(score_batch_device(model, rgb, op, s, e, None, exact=True).cpu().numpy(), e - s)
                _count_fallback(model)
                reruns += 1
            scores.append(sc)
        rec = assemble_records(rgb.shape[0], batches, scores)
        out["rgb_img_pred_records"].append(rec["rgb_psnr"])
        out["rgb_fea_comm_records"].append(rec["rgb_comm"])
        out["op_img_pred_records"].append(rec["op_psnr"])
        out["op_fea_comm_records"].append(rec["op_comm"])

    pending = None
    for rgb, op in subvideos:
        if op.shape[0] != rgb.shape[0] - 1:
            raise ValueError("a sub-video of T frames has T-1 flows")
        batches = subvideo_batches(rgb.shape[0])
        flat = torch.cat([score_batch_device(model, rgb, op, s, e, None) for s, e in batches])
        if flat.is_cuda:
            host = torch.empty(flat.numel(), dtype=flat.dtype, pin_memory=True)
            host.copy_(flat, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            copies += 1
        else:
            host, ev = flat, _Done()
        if pending is not None:
            finish(pending)
        pending = (rgb, op, batches, host, ev)
    if pending is not None:
        finish(pending)
    if stats is not None:
        stats.update(score_copies=copies, rerun_batches=reruns)
    return out


class _Done:
    def synchronize(self):
        pass


def generator_loss(out, rgb_t: torch.Tensor, op_t: torch.Tensor, lam_lp: float = 1.0, lam_lp_op: float = 1.0,
                   lam_latent: float = 1.0) -> torch.Tensor:
    rgb, op, (rd, od), _ = out[:4]
    l_rgb = torch.norm(rgb - rgb_t, p=2, dim=1).mean()            # `L2`, losses_utils.py:124-129
    l_op = torch.norm(op - op_t, p=2, dim=1).mean()
    return lam_lp * l_rgb + lam_lp_op * l_op + lam_latent * (rd + od).sum()


def adam(params, lr: float, **kw) -> torch.optim.Adam:
    """the reference's optimizer (`torch.optim.Adam(params, lr=...)`, optimizer/__init__.py:39-49) - the same update
    rule in torch's FUSED form when the parameters live on the GPU: one multi-tensor launch per ~chunk of parameters
    instead of the ~17 elementwise launches of the default foreach form (0.48 -> ~0.15 ms per batch-32 step)"""
    params = list(params)
    fused = bool(params) and all(p.is_cuda for p in params)
    return torch.optim.Adam(params, lr=lr, fused=fused, **kw)


def train_step(model: torch.nn.Module, optimizer: torch.optim.Optimizer, rgb: torch.Tensor, op: torch.Tensor,
               **lams) -> torch.Tensor:
    """one G step of the joint training loop (train_helper.py:296-339 without D / FlowNet terms).
    rgb [B,5,3,H,W], op [B,4,2,H,W]; targets are the last frame of each.  With a
    `parallel.BucketedGradReducer` attached to the model the gradients are averaged across ranks
    inside backward (RCCL over xGMI)."""
    b = rgb.shape[0]
    rgb_in = rgb[:, :-1].reshape(b, -1, *rgb.shape[-2:])
    op_in = op[:, :-1].reshape(b, -1, *op.shape[-2:])
    optimizer.zero_grad(set_to_none=True)
    loss = generator_loss(model(rgb_in, op_in), rgb[:, -1], op[:, -1], **lams)
    vote, group = _watch_group(model)
    watch = _FiniteWatch(loss, group=group, vote=vote)
    loss.backward()
    watch.step(optimizer)
    return loss.detach()


def _watch_group(*models):
    """(vote, group) for `_FiniteWatch`: ranks share the verdict on the loss exactly when their gradients or statistics
    are shared - a `parallel.BucketedGradReducer` or `parallel.sync_statistics` attached to one of the models - and in
    THAT process group.  Models that train on their own (no reducer: per-rank runs inside an initialised world, the
    stress-style harnesses) do not vote: no rank waits in a collective the others never enter."""
    for m in models:
        red = getattr(m, "_grad_reducer", None)
        if red is not None:
            return True, getattr(red, "group", None)
        sync = getattr(m, "_sync_stats", None)
        if sync and sync[0]:
            return True, sync[1]
    return False, None


class _FiniteWatch:
    """`optimizer.step()` unless the loss is non-finite, without stalling the device.  The split-fp16 training kernels
    (`train_precision = "s16"`, the default) encode their operands in half range: an activation beyond 65504 becomes inf
    in the re-encoding and reaches the loss as inf / NaN; stepping would write that into the Adam moments and the
    weights for good.  The verdict on the loss is copied to pinned host memory right after the FORWARD (one 1-byte copy
    + an event, queued before the backward's kernels); `step` waits for that event only - by then the host has enqueued
    the whole backward, so the device keeps working while the host waits - and refuses the step loudly.  (The
    BatchNorm / EMA buffers of the refused step were updated in place by its forward, as the reference's would be.)
    Data parallel: the gradients are averaged across ranks INSIDE backward, so one rank's non-finite loss poisons every
    rank's gradients; the verdict is therefore all-reduced (MIN) on the device before it is copied, and all ranks refuse
    the step together (no rank steps on NaN gradients, none is left waiting in the next collective)."""

    def __init__(self, *losses, group=None, vote: bool = True, flags=()):
        """`vote` / `group`: whether the verdict is shared with other ranks, and in which process group - the group of
        the model's gradient reducer / synchronised statistics (`_watch_group`): ranks outside it, or ranks that train
        independently inside an initialised world, never enter this collective, so it must not run on the default group."""
        fin = torch.stack([torch.isfinite(v.detach()).all() for v in losses] +
                          [(f.reshape(-1)[0] == 0) for f in flags if f is not None])       # S16 range flags of frozen / side networks
        self.flag = torch.empty(fin.numel(), dtype=torch.bool).pin_memory() if losses[0].is_cuda else None
        if vote and parallel.dist.is_available() and parallel.dist.is_initialized() and parallel.dist.get_world_size(group) > 1:
            vote_t = fin.to(torch.int32)
            parallel.dist.all_reduce(vote_t, op=parallel.dist.ReduceOp.MIN, group=group)
            fin = vote_t.to(torch.bool)
        if self.flag is not None:
            self.flag.copy_(fin, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.flag, self.event = fin, None

    def step(self, optimizer) -> None:
        if self.event is not None:
            self.event.synchronize()
        if not bool(self.flag.all()):
            raise FloatingPointError(
                "non-finite training loss: an operand left the fp16 range of the split-fp16 kernels (or the run diverged); "
                "optimizer.step() was NOT taken - set model.train_precision = 'fp32' to train this step on the exact kernels")
        optimizer.step()


# ---- the remaining loss terms and the alternating G / D step -----------------------------------

LAMS_ANOPRED = dict(lam_adv=0.05, lam_gdl=1.0, lam_flow=2.0, lam_lp=1.0, lam_lp_op=1.0, lam_latent=1.0)
"""the reference reads its lambdas from per-dataset .ini files that are not shipped (constant_train.py:282-292);
these are the values of the ano_pred recipe the training loop was taken from"""


def gradient_loss(gen: torch.Tensor, gt: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """`Gradient_Loss` (losses_utils.py:30-61): the [-1,1] filters sum over channels, zero pad on the left / top"""
    def dxy(t):
        s = t.sum(1, keepdim=True)
        dx = torch.cat([s[..., :1], s[..., 1:] - s[..., :-1]], dim=-1)
        dy = torch.cat([s[..., :1, :], s[..., 1:, :] - s[..., :-1, :]], dim=-2)
        return dx, dy
    gx, gy = dxy(gen)
    tx, ty = dxy(gt)
    ex, ey = (tx - gx).abs(), (ty - gy).abs()
    if alpha != 1:
        ex, ey = ex ** alpha, ey ** alpha
    return (ex + ey).mean()


def adversarial_loss(fake_outputs: torch.Tensor) -> torch.Tensor:
    """`Adversarial_Loss` (losses_utils.py:100-104)"""
    return ((fake_outputs - 1) ** 2 / 2).mean()


def discriminate_loss(real_outputs: torch.Tensor, fake_outputs: torch.Tensor) -> torch.Tensor:
    """`Discriminate_Loss` (losses_utils.py:106-110)"""
    return ((real_outputs - 1) ** 2 / 2).mean() + (fake_outputs ** 2 / 2).mean()


def flow_loss(gen_flows: torch.Tensor, gt_flows: torch.Tensor) -> torch.Tensor:
    """`Flow_Loss` (losses_utils.py:10-15)"""
    return (gen_flows - gt_flows).abs().mean()


def generator_loss_full(out, rgb_t, op_t, d_gen, flow_pred=None, flow_gt=None, lam_adv=0.05, lam_gdl=1.0, lam_flow=2.0,
                        lam_lp=1.0, lam_lp_op=1.0, lam_latent=1.0) -> torch.Tensor:
    """`Twostream_vq_Loss.forward` (loss_zoo.py:310-336); the FlowNet2-SD term enters only through precomputed
    flows"""
    rgb = out[0]
    loss = generator_loss(out, rgb_t, op_t, lam_lp, lam_lp_op, lam_latent) + lam_adv * adversarial_loss(d_gen) + \
        lam_gdl * gradient_loss(rgb, rgb_t)
    if flow_pred is not None:
        loss = loss + lam_flow * flow_loss(flow_pred, flow_gt)
    return loss


def train_step_gan(generator: torch.nn.Module, discriminator: torch.nn.Module, optimizer_G, optimizer_D,
                   rgb: torch.Tensor, op: torch.Tensor, flow_fn: Optional[Callable] = None, **lams):
    """One iteration of the joint loop (train_helper.py:296-339): G forward, D(G(x)) for the adversarial term, the D
    update on (target, detached prediction), then the G update.  The gradient that reaches G through D uses the
    filters D had when `d_gen` was computed (before its update), i.e. the exact derivative of the loss value.
    `flow_fn(prev_frame, frame) -> flow` stands in for FlowNet2-SD when given."""
    b = rgb.shape[0]
    rgb_in = rgb[:, :-1].reshape(b, -1, *rgb.shape[-2:])
    op_in = op[:, :-1].reshape(b, -1, *op.shape[-2:])
    rgb_t, op_t = rgb[:, -1], op[:, -1]
    out = generator(rgb_in, op_in)
    vote, group = _watch_group(generator, discriminator)
    # Round 6 (`AMMC_GAN_OVERLAP`, default on): the D update is independent of everything the generator still has to do - it
    # reads the prediction detached, and the gradient that reaches G through D uses the filter packs of the `d_gen`
    # forward, not the live parameters - so its forward (beside FlowNet2-SD and the `d_gen` forward), its backward and its
    # Adam step (beside the generator's backward, whose BatchNorm passes leave the matrix pipe idle: the two-stream argument
    # of DESIGN.md section 4) run on a SECOND HIP stream.  Same kernels, same order on each stream: every number is what
    # the serial form computes.  Off with a gradient reducer / synchronised statistics attached (their collectives are
    # ordered on the caller's stream) and on the CPU.
    overlap = GAN_OVERLAP and out[0].is_cuda and not vote and getattr(discriminator, "_grad_reducer", None) is None
    main = torch.cuda.current_stream(out[0].device) if overlap else None
    lane = _gan_side_stream(out[0].device) if overlap else None

    def d_forward():
        # D(real) and D(fake.detach()) (train_helper.py:326-327) as one call on 2 b frames: the discriminator has no
        # batch-coupled layer, the patch maps and every gradient are those of the two calls
        d_both_ = discriminator(torch.cat([rgb_t, out[0].detach()]))
        return d_both_, discriminate_loss(d_both_[:b], d_both_[b:]), getattr(discriminator, "last_overflow", None)
    early = overlap and GAN_OVERLAP_EARLY
    if early:
        lane.wait_stream(main)
        with torch.cuda.stream(lane):
            d_both, d_loss, d_flag = d_forward()
    flow_pred = flow_gt = None
    flow_mods = [m for m in (getattr(flow_fn, "__self__", None), getattr(flow_fn, "net", None)) if m is not None]
    if flow_fn is not None:
        with torch.no_grad():
            # the reference pairs BOTH frames with `rgb_input_last = rgb[:, -1]`, i.e. with the target frame itself
            # (train_helper.py:299, 309-312), not with the frame before it: followed as written.  The two FlowNet2-SD
            # forwards run as ONE batch of 2 b pairs (every sample is independent in that network - its mean subtraction
            # is per sample - so the flows are the same; the layers below 1/16 resolution fill the chip twice as well)
            both = flow_fn(torch.cat([rgb[:, -1], rgb[:, -1]]), torch.cat([out[0].detach(), rgb_t]))
            flow_pred, flow_gt = both[:b], both[b:]
    d_params = [p for p in discriminator.parameters() if p.requires_grad]
    for p in d_params:                 # the G step needs dL/d(frame) through D, not D's weight gradients
        p.requires_grad_(False)
    try:
        d_gen = discriminator(out[0])
    finally:
        for p in d_params:
            p.requires_grad_(True)
    g_loss = generator_loss_full(out, rgb_t, op_t, d_gen, flow_pred, flow_gt, **lams)
    if not early:
        if overlap:
            lane.wait_stream(main)
        with (torch.cuda.stream(lane) if overlap else contextlib.nullcontext()):
            d_both, d_loss, d_flag = d_forward()
    if overlap:
        main.wait_stream(lane)                     # (the verdict below reads d_loss on the caller's stream)
        for t in (d_both, d_loss) + ((d_flag,) if d_flag is not None else ()):
            t.record_stream(main)
    # (the S16 range flags of the discriminator and - where `flow_fn` is a bound method / has `.net` - of the frozen flow
    # estimator join the verdict: `s16_guard = "defer"` on those modules leaves them on the device instead of syncing)
    watch = _FiniteWatch(d_loss, g_loss, group=group, vote=vote,
                         flags=[d_flag] + [getattr(m, "last_overflow", None) for m in flow_mods])
    with (torch.cuda.stream(lane) if overlap else contextlib.nullcontext()):
        optimizer_D.zero_grad(set_to_none=True)
        d_loss.backward()                          # (autograd runs it on the stream of its forward: the lane)
        watch.step(optimizer_D)
    optimizer_G.zero_grad(set_to_none=True)
    g_loss.backward()
    optimizer_G.step()
    if overlap:
        main.wait_stream(lane)                     # successors on the caller's stream see both updates
    return g_loss.detach(), d_loss.detach()


# (the two FlowNet2-SD forwards on a third stream, their no-gradient term joined to the loss value at the end, were built and
# measured as well: 80.2 / 80.4 ms against 80.2 / 79.3 - nothing; removed)
# AMMC_GAN_OVERLAP: 2 (default) = the D lane starts right behind the generator's forward (its own forward beside FlowNet2-SD and
# the d_gen forward: 79.84 / 80.20 / 79.85 -> 79.04 / 78.82 / 78.88 ms against 1), 1 = it starts behind g_loss (the first form:
# 81.69 / 81.59 -> 78.94 / 79.19 against 0), 0 = serial
GAN_OVERLAP = os.environ.get("AMMC_GAN_OVERLAP", "2") != "0"
GAN_OVERLAP_EARLY = os.environ.get("AMMC_GAN_OVERLAP", "2") == "2"
_GAN_LANES: Dict = {}


def _gan_side_stream(device, which: int = 0) -> "torch.cuda.Stream":
    key = (device.type, device.index, which)
    if key not in _GAN_LANES:
        _GAN_LANES[key] = torch.cuda.Stream(device=device)
    return _GAN_LANES[key]


# ---- score fusion and frame-level AUC (the step after the records) ------------------------------

LAM_MAP = {"avenue": (0.04, 0.65), "ped2": (0.01, 0.55), "shanghaitech": (0.13, 0.60)}   # test_helper.py:565-569
DECIDABLE_IDX = 4                                                                          # eval_metric.py:17


def _minmax_concat(records: Sequence[np.ndarray]) -> np.ndarray:
    """per-video min-max normalisation, drop the first DECIDABLE_IDX frames, global min-max
    (`norm_score`, main/eval_metric.py:405-417)"""
    parts = []
    for r in records:
        r = np.array(r, dtype=np.float32)
        r = r - r.min()
        r = r / r.max()
        parts.append(r[DECIDABLE_IDX:])
    s = np.concatenate(parts)
    s = s - s.min()
    return s / s.max()


def roc_auc(labels: np.ndarray, scores: np.ndarray, pos_label: int = 0) -> float:
    """area under the ROC curve by the trapezoid rule over distinct thresholds (what
    sklearn.metrics.roc_curve + auc compute; ties share one threshold)"""
    y = (np.asarray(labels) == pos_label)
    s = np.asarray(scores, dtype=np.float64)
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[idx].astype(np.float64)
    fps = (1 + idx - tps).astype(np.float64)
    tps, fps = np.r_[0.0, tps], np.r_[0.0, fps]
    trapezoid = getattr(np, "trapezoid", None) or np.trapz
    return float(trapezoid(tps / tps[-1], fps / fps[-1]))


def fuse_scores_auc(records: dict, gt: Sequence[np.ndarray], lam=None) -> dict:
    """`img_pred_fea_comm_single_auc` (main/eval_metric.py:382-439): normality score
    (1-l1)*psnr_n + l1*(1 - commit_n), one-tap smoothing s'_i = (1-l2) s_{i-1} + l2 s_i (from the
    UNSMOOTHED neighbour), ROC with the normal class (label 0) positive, AUC rounded to 3 d.p.
    `records` is the dict of `evaluate_dataset`; `gt[i]` the per-frame labels of video i (1 = anomalous)."""
    l1, l2 = lam if lam is not None else LAM_MAP[records["dataset"]]
    labels = np.concatenate([np.asarray(g)[DECIDABLE_IDX:] for g in gt])
    img = _minmax_concat(records["rgb_img_pred_records"])
    fea = _minmax_concat(records["rgb_fea_comm_records"])
    scores = (1 - l1) * img + l1 * (1.0 - fea)
    smooth = scores.copy()
    smooth[1:] = (1 - l2) * scores[:-1] + l2 * scores[1:]
    auc = roc_auc(labels, smooth, pos_label=0)
    return {"auc": round(auc, 3), "auc_raw": auc, "lam": (l1, l2), "scores": smooth}


def weights_init_normal(model: torch.nn.Module) -> None:
    """the reference's training-from-scratch init (utils/utils.py:328-334): Conv* ~ N(0, 0.02),
    BatchNorm2d gamma ~ N(1, 0.02), beta = 0; applied by class name exactly as there"""
    for m in model.modules():
        cn = m.__class__.__name__
        if cn.find("Conv") != -1 and hasattr(m, "weight") and isinstance(m.weight, torch.nn.Parameter):
            torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
        elif cn.find("BatchNorm2d") != -1:
            torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
            torch.nn.init.constant_(m.bias.data, 0.0)
    if hasattr(model, "_param_epoch"):
        model._param_epoch += 1


# ---- checkpoint tooling (utils/utils.py:182-263): same file names and key conventions as the reference ------------

def save_checkpoint(state_dict: dict, model_dir: str, step: int) -> str:
    """`saver` (utils.py:182-189): <model_dir>/step_<step+1, 6 digits>.pth"""
    import os
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, "step_{}.pth".format(str(step + 1).zfill(6)))
    torch.save(state_dict, path)
    return path


def load_latest_checkpoint(model: torch.nn.Module, model_dir: str, map_location="cpu"):
    """`loader` (utils.py:192-203): the lexicographically last file of the directory; returns (model, step)"""
    import os
    names = sorted(os.listdir(model_dir))
    if not names:
        raise FileNotFoundError(f"no checkpoint in {model_dir}")
    path = os.path.join(model_dir, names[-1])
    model.load_state_dict(torch.load(path, map_location=map_location))
    return model, int(path.split("_")[-1].split(".")[0])


def load_pretrained_branches(model: torch.nn.Module, rgb_ckpt, op_ckpt, map_location="cpu"):
    """`loader_rgb_op_branch` (utils.py:236-263): the two-stage recipe of the reference README - single-stream
    `UNetMem_v7` checkpoints (paths or state dicts) go into the `rgb.` / `op.` branches of `twostream`, keys the joint
    model does not have are dropped, the bridge keeps its initialisation.  Returns (model, 0) like the reference."""
    def as_dict(c):
        return torch.load(c, map_location=map_location) if isinstance(c, (str, bytes)) or hasattr(c, "__fspath__") else c
    own = model.state_dict()
    for prefix, ck in (("rgb", as_dict(rgb_ckpt)), ("op", as_dict(op_ckpt))):
        own.update({f"{prefix}.{k}": v for k, v in ck.items() if f"{prefix}.{k}" in own})
    model.load_state_dict(own)
    return model, 0


def flownet_flow_fn(flownet: torch.nn.Module) -> Callable:
    """`flow_fn` for `train_step_gan` from a (frozen, eval-mode) FlowNet2-SD: frames in [-1, 1] -> flow / 255, as
    train_helper.py:309-316 feeds it (`(pair * 0.5 + 0.5) * 255`, then `/ 255`, detached)"""
    def fn(prev: torch.Tensor, cur: torch.Tensor) -> torch.Tensor:
        pair = torch.cat([prev.unsqueeze(2), cur.unsqueeze(2)], 2)
        with torch.no_grad():
            return flownet((pair * 0.5 + 0.5) * 255.0) / 255.0
    return fn
