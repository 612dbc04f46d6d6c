"""CPU oracle for the appearance-motion memory-consistency path.  TEST INFRASTRUCTURE.

This file restates, in plain functional PyTorch CPU ops over a flat
`state_dict`, the arithmetic of the reference's hot path.  It is the checker
that the HIP path is compared against; it is never the product.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference's
`Code/models/unet.py` (authoring container only) and records its outputs on
deterministic inputs; `tests/test_oracle_golden.py` checks this restatement
against those committed vectors (<= 1e-6 relative) and, when /root/reference is
present, against the live reference module.  Dataset-level AUC (README
screenshots) is unpinned: no checkpoints or data exist.

Each function cites the reference lines it follows (paths relative to
/root/reference/Code).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

BN_EPS = 1e-5          # nn.BatchNorm2d default, models/unet.py:12,15
BN_MOMENTUM = 0.1
VQ_DECAY = 0.99        # models/unet.py:268
VQ_EPS = 1e-5


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------

def _bn(sd: State, p: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """nn.BatchNorm2d (models/unet.py:12,15): batch stats + running update in
    training, running stats in eval."""
    if training:
        sd[f"{p}.num_batches_tracked"] += 1
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"],
                        sd[f"{p}.weight"], sd[f"{p}.bias"], training, BN_MOMENTUM, BN_EPS)


def double_conv(sd: State, p: str, x: torch.Tensor, training: bool = False) -> torch.Tensor:
    """[conv3x3 p1 no-bias -> BN -> ReLU] x2   (models/unet.py:8-20)."""
    x = F.conv2d(x, sd[f"{p}.0.weight"], None, padding=1)
    x = F.relu(_bn(sd, f"{p}.1", x, training))
    x = F.conv2d(x, sd[f"{p}.3.weight"], None, padding=1)
    x = F.relu(_bn(sd, f"{p}.4", x, training))
    return x


def maxpool2x2_routes(x: torch.Tensor):
    """MaxPool2d(2) (floor mode) and WHICH element of each window it took: (pooled, routes [B,C,h,w] int64 in 0..3, window
    position row-major, the first maximum as F.max_pool2d takes it)"""
    b, c, hh, ww = x.shape
    y, flat = F.max_pool2d(x, 2, return_indices=True)
    iy, ix = flat // ww, flat % ww
    return y, (iy & 1) * 2 + (ix & 1)


def maxpool2x2_forced(x: torch.Tensor, routes: torch.Tensor) -> torch.Tensor:
    """TEST INSTRUMENTATION (no counterpart in the reference, like `quantize_topk(force_idx=...)`): the pooling with its
    routes prescribed - element `routes[b,c,i,j]` (0..3, row-major) of window (i, j).  A 2x2 window whose two largest
    values tie within rounding noise is the path's second discontinuity: which one an fp32-accurate evaluation takes is
    noise, and the gradient follows it to another pixel.  The fp64 truth of tests/truth.py takes the routes of the
    evaluation under test where that evaluation records them (the HIP training engine does: one byte per pooled element)."""
    b, c, hh, ww = x.shape
    h, w = hh // 2, ww // 2
    win = x[..., :2 * h, :2 * w].reshape(b, c, h, 2, w, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, h, w, 4)
    return win.gather(-1, routes.to(device=x.device, dtype=torch.int64).unsqueeze(-1)).squeeze(-1)


def down(sd: State, p: str, x: torch.Tensor, training: bool = False, force_pool: torch.Tensor = None, aux: dict = None) -> torch.Tensor:
    """MaxPool2d(2) then double_conv   (models/unet.py:33-41).  `force_pool` / `aux`: test instrumentation - the pooling
    routes to take (`maxpool2x2_forced`) / a dict that receives the routes this evaluation took under `"<p>.pool"`."""
    if force_pool is not None:
        pooled = maxpool2x2_forced(x, force_pool)
    elif aux is not None:
        pooled, aux[f"{p}.pool"] = maxpool2x2_routes(x)
    else:
        pooled = F.max_pool2d(x, 2)
    return double_conv(sd, f"{p}.mpconv.1.conv", pooled, training)


def up(sd: State, p: str, x1: torch.Tensor, x2: torch.Tensor, training: bool = False) -> torch.Tensor:
    """ConvTranspose2d(C, C/2, 2, stride 2) + bias, pad to the skip's size,
    cat([skip, upsampled]), double_conv   (models/unet.py:44-59)."""
    x1 = F.conv_transpose2d(x1, sd[f"{p}.up.weight"], sd[f"{p}.up.bias"], stride=2)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    return double_conv(sd, f"{p}.conv.conv", torch.cat([x2, x1], dim=1), training)


def quantize_topk(x: torch.Tensor, embed: torch.Tensor, k: int, force_idx: torch.Tensor = None):
    """Memory addressing, forward arithmetic only (models/unet.py:282-297, 310).

    x [B,h,w,D], embed [D,M].  Returns (q_topk [B,h,w,k*D], diff scalar,
    idx_topk [B,h,w,k] int64, idx_top1 [N] int64, flatten [N,D]).
    Same expression order as the reference: |x|^2 - 2 x.E + |E|^2.

    `force_idx` [N,k] (TEST INSTRUMENTATION, no counterpart in the reference): take these slots instead of the ranking's.
    The lookup is the path's one hard discontinuity - two slots whose distances tie within rounding noise - so two
    evaluations of a training step can only be compared entry by entry on the SAME branch of it
    (tests/test_gpu_train.py: the fp64 truth is evaluated with the lookups of the evaluation under test).
    """
    d = embed.shape[0]
    flatten = x.reshape(-1, d)
    dist = (flatten.pow(2).sum(1, keepdim=True)
            - 2 * flatten @ embed
            + embed.pow(2).sum(0, keepdim=True))
    idx1 = (-dist).max(1)[1]
    table = embed.transpose(0, 1)
    if force_idx is not None:
        forced = force_idx.to(device=x.device, dtype=torch.int64).reshape(-1, k)
        idx1 = forced[:, 0].contiguous()
    q1 = F.embedding(idx1.view(*x.shape[:-1]), table)
    idxk = (-dist).topk(k, dim=1)[1].view(x.shape[0], x.shape[1], x.shape[2], -1)
    if force_idx is not None:
        idxk = forced.view(x.shape[0], x.shape[1], x.shape[2], -1)
    qk = F.embedding(idxk, table).view(x.shape[0], x.shape[1], x.shape[2], -1)
    diff = (q1.detach() - x).pow(2).mean()
    return qk, diff, idxk, idx1, flatten, q1


def codebook_ema_update(sd: State, p: str, flatten: torch.Tensor, idx1: torch.Tensor) -> None:
    """EMA codebook update, in place on the buffers (models/unet.py:298-309)."""
    embed = sd[f"{p}.embed"]
    m = embed.shape[1]
    onehot = F.one_hot(idx1, m).type(flatten.dtype)
    with torch.no_grad():
        cs = sd[f"{p}.cluster_size"]
        ea = sd[f"{p}.embed_avg"]
        cs.mul_(VQ_DECAY).add_(onehot.sum(0), alpha=1 - VQ_DECAY)
        ea.mul_(VQ_DECAY).add_(flatten.detach().transpose(0, 1) @ onehot, alpha=1 - VQ_DECAY)
        n = cs.sum()
        smoothed = (cs + VQ_EPS) / (n + m * VQ_EPS) * n
        embed.copy_(ea / smoothed.unsqueeze(0))


def vq_block(sd: State, p: str, x: torch.Tensor, k: int, training: bool = False, force_idx: torch.Tensor = None):
    """enc 1x1 -> NHWC -> memory read -> NCHW -> dec 1x1 -> += x
    (models/unet.py:318-331 and 379-387).  Returns (out, diff[1], q_one, idx_topk)."""
    q = f"{p}.quan"
    z = F.conv2d(x, sd[f"{q}.enc.weight"], sd[f"{q}.enc.bias"]).permute(0, 2, 3, 1)
    qk, diff, idxk, idx1, flatten, q1 = quantize_topk(z, sd[f"{q}.quantize.embed"], k, force_idx)
    if training:
        codebook_ema_update(sd, f"{q}.quantize", flatten, idx1)
    q_one = z + (q1 - z).detach()                                   # unet.py:311
    out = F.conv2d(qk.permute(0, 3, 1, 2), sd[f"{q}.dec.weight"], sd[f"{q}.dec.bias"])
    out = out + x                                                   # unet.py:386
    return out, diff.unsqueeze(0), q_one, idxk


def bridge(sd: State, zx: torch.Tensor, zy: torch.Tensor, training: bool = False):
    """AMFT: x = zx + O2F(zy); y = zy + F20(zx)   (models/unet.py:956-965)."""
    x = zx + double_conv(sd, "bridge.O2F.conv", zy, training)
    y = zy + double_conv(sd, "bridge.F20.conv", zx, training)
    return x, y


# ----------------------------------------------------------------------------
# whole models
# ----------------------------------------------------------------------------

def unet_forward(sd: State, x: torch.Tensor, training: bool = False, prefix: str = "") -> torch.Tensor:
    """`UNet.forward` (models/unet.py:73-83): config 1 of BASELINE.json."""
    p = prefix
    x1 = double_conv(sd, f"{p}inc.conv.conv", x, training)
    x2 = down(sd, f"{p}down1", x1, training)
    x3 = down(sd, f"{p}down2", x2, training)
    x4 = down(sd, f"{p}down3", x3, training)
    y = up(sd, f"{p}up1", x4, x3, training)
    y = up(sd, f"{p}up2", y, x2, training)
    y = up(sd, f"{p}up3", y, x1, training)
    y = F.conv2d(y, sd[f"{p}outc.weight"], sd[f"{p}outc.bias"], padding=1)
    return torch.tanh(y)


def unetmem_forward(sd: State, x: torch.Tensor, k: int, training: bool = False, prefix: str = "",
                    force_idx: torch.Tensor = None, want_idx: bool = False):
    """`UNetMem_v7.forward` (models/unet.py:924-937).  `force_idx` / `want_idx`: test instrumentation, see `quantize_topk`
    (with `want_idx` the lookups [N, k] are returned as a fourth value)."""
    p = prefix
    x1 = double_conv(sd, f"{p}inc.conv.conv", x, training)
    x2 = down(sd, f"{p}down1", x1, training)
    x3 = down(sd, f"{p}down2", x2, training)
    x4 = down(sd, f"{p}down3", x3, training)
    x4, diff, q_one, idxk = vq_block(sd, f"{p}vq_down3", x4, k, training, force_idx)
    y = up(sd, f"{p}up1", x4, x3, training)
    y = up(sd, f"{p}up2", y, x2, training)
    y = up(sd, f"{p}up3", y, x1, training)
    y = F.conv2d(y, sd[f"{p}outc.weight"], sd[f"{p}outc.bias"], padding=1)
    if want_idx:
        return torch.tanh(y), diff, q_one, idxk.reshape(-1, k).detach()
    return torch.tanh(y), diff, q_one


def twostream_forward(sd: State, rgb_x: torch.Tensor, op_x: torch.Tensor, k: int = 2,
                      training: bool = False, want_aux: bool = False, force_idx: dict = None):
    """`twostream.forward` (models/unet.py:981-1007), same operation order
    (it matters in training: BN running stats and EMA buffers are updated as
    each submodule runs).

    Returns (rgb, op, (rgb_diff[1], op_diff[1]), (rgb_q, op_q)) and, if
    `want_aux`, a dict of intermediates for per-stage parity checks.
    """
    aux = {}

    fi = force_idx or {}                  # test instrumentation: {"rgb": idx [N, k], "op": ..., "pool": {"rgb.down1": routes, ...}}
    fpool = fi.get("pool") or {}
    pools = {} if want_aux else None

    def enc(p, x):
        x1 = double_conv(sd, f"{p}.inc.conv.conv", x, training)
        x2 = down(sd, f"{p}.down1", x1, training, fpool.get(f"{p}.down1"), pools)
        x3 = down(sd, f"{p}.down2", x2, training, fpool.get(f"{p}.down2"), pools)
        x4 = down(sd, f"{p}.down3", x3, training, fpool.get(f"{p}.down3"), pools)
        return x1, x2, x3, x4

    def dec(p, x4, x3, x2, x1):
        y = up(sd, f"{p}.up1", x4, x3, training)
        aux[f"{p}.u1"] = y
        y = up(sd, f"{p}.up2", y, x2, training)
        aux[f"{p}.u2"] = y
        y = up(sd, f"{p}.up3", y, x1, training)
        aux[f"{p}.u3"] = y
        return F.conv2d(y, sd[f"{p}.outc.weight"], sd[f"{p}.outc.bias"], padding=1)

    r1, r2, r3, r4 = enc("rgb", rgb_x)
    aux.update({"rgb.x1": r1, "rgb.x2": r2, "rgb.x3": r3, "rgb.x4": r4})
    r4q, rgb_diff, rgb_q, rgb_idx = vq_block(sd, "rgb.vq_down3", r4, k, training, fi.get("rgb"))
    o1, o2, o3, o4 = enc("op", op_x)
    aux.update({"op.x1": o1, "op.x2": o2, "op.x3": o3, "op.x4": o4})
    o4q, op_diff, op_q, op_idx = vq_block(sd, "op.vq_down3", o4, k, training, fi.get("op"))
    aux.update({"rgb.vq": r4q, "op.vq": o4q, "rgb.idx": rgb_idx, "op.idx": op_idx})
    r4b, o4b = bridge(sd, r4q, o4q, training)
    aux.update({"rgb.bridge": r4b, "op.bridge": o4b})
    rgb = dec("rgb", r4b, r3, r2, r1)
    op = dec("op", o4b, o3, o2, o1)
    out = (torch.tanh(rgb), torch.tanh(op), (rgb_diff, op_diff), (rgb_q, op_q))
    if want_aux:
        # the routes actually taken: the forced ones where given, else this evaluation's own ("<stream>.down<i>.pool" -> "<stream>.down<i>")
        aux["pool"] = {**{k[:-5]: v for k, v in pools.items()}, **{k: v for k, v in fpool.items()}}
    return out + (aux,) if want_aux else out


# ----------------------------------------------------------------------------
# the steps either side of the path: scoring (eval) and the loss (train)
# ----------------------------------------------------------------------------

def psnr_error(gen: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:130-148 — mean over the batch of per-sample PSNR on [0,1]."""
    n = gen.shape[1] * gen.shape[2] * gen.shape[3]
    sq = ((gt + 1.0) / 2.0 - (gen + 1.0) / 2.0) ** 2
    per = 10.0 * torch.log10(1.0 / ((1.0 / n) * sq.sum(dim=[1, 2, 3])))
    return per.mean()


def eval_subvideo_records(forward, rgb_frames: torch.Tensor, op_frames: torch.Tensor,
                          rgb_len_clip: int = 5, op_len_clip: int = 4, batch: int = 16):
    """Per-frame records of one sub-video, with the reference loop's semantics
    (run_helper/test_helper.py:408-473): sliding clips in order, batches of 16
    (last short), PSNR per sample, the batch's commit score written to every
    frame of the batch, the first len_clip-1 frames back-filled, and the op
    arrays' last entry copied from the one before it.

    rgb_frames [T,3,H,W]; op_frames [T-1,2,H,W] (T-1 flows for T frames).
    `forward(rgb_in, op_in)` returns the model's 4-tuple.  The op "target" in
    the reference is shape-mismatched and its PSNR unused by the AUC
    (test_helper.py:431); we score op against the last input flow instead and
    do not pin it.
    """
    import numpy as np

    t = rgb_frames.shape[0]
    n_clip = t - rgb_len_clip + 1
    # T frames have T-1 flows, so both datasets yield the same number of clips
    assert op_frames.shape[0] - op_len_clip + 1 == n_clip
    rec = {key: np.empty((t,), dtype=np.float32)
           for key in ("rgb_psnr", "rgb_comm", "op_psnr", "op_comm")}
    cnt = -1
    for s in range(0, n_clip, batch):
        e = min(s + batch, n_clip)
        b = e - s
        rgb = torch.stack([rgb_frames[i:i + rgb_len_clip] for i in range(s, e)])
        op = torch.stack([op_frames[i:i + op_len_clip] for i in range(s, e)])
        rgb_in = rgb[:, :-1].reshape(b, -1, *rgb.shape[-2:])
        rgb_t = rgb[:, -1]
        op_in = op[:, :-1].reshape(b, -1, *op.shape[-2:])
        with torch.no_grad():
            rgb_out, op_out, (rgb_diff, op_diff), _ = forward(rgb_in, op_in)
        for i in range(b):
            cnt += 1
            rec["rgb_psnr"][cnt + rgb_len_clip - 1] = float(psnr_error(rgb_out[i:i + 1], rgb_t[i:i + 1]))
            rec["rgb_comm"][cnt + rgb_len_clip - 1] = float(rgb_diff)
            rec["op_psnr"][cnt + op_len_clip - 1] = float(psnr_error(op_out[i:i + 1], op[i:i + 1, -1]))
            rec["op_comm"][cnt + op_len_clip - 1] = float(op_diff)
    for key, lc in (("rgb_psnr", rgb_len_clip), ("rgb_comm", rgb_len_clip),
                    ("op_psnr", op_len_clip), ("op_comm", op_len_clip)):
        rec[key][:lc - 1] = rec[key][lc - 1]
    rec["op_psnr"][t - 1] = rec["op_psnr"][t - 2]
    rec["op_comm"][t - 1] = rec["op_comm"][t - 2]
    return rec


def intensity_l2(gen: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """`L2` (models/losses/losses_utils.py:124-129): mean per-pixel channel norm."""
    return torch.norm(gen - gt, p=2, dim=1).mean()


def generator_loss(out, rgb_t: torch.Tensor, op_t: torch.Tensor,
                   lam_lp: float = 1.0, lam_lp_op: float = 1.0, lam_latent: float = 1.0) -> torch.Tensor:
    """The kernel-parity subset of `Twostream_vq_Loss` (loss_zoo.py:323-336):
    lam_lp*L2(rgb) + lam_lp_op*L2(op) + lam_latent*(rgb_diff + op_diff).  The
    reference multiplies a *tuple* by lam_latent (a TypeError); the ablation
    twin sums the two (unet.py:1065), which is the evident intent."""
    rgb, op, (rd, od), _ = out[:4]
    return lam_lp * intensity_l2(rgb, rgb_t) + lam_lp_op * intensity_l2(op, op_t) + \
        lam_latent * (rd + od).sum()


# ---- PixelDiscriminator and the remaining generator / discriminator loss terms -------------

def pixel_discriminator(sd: State, x: torch.Tensor) -> torch.Tensor:
    """`PixelDiscriminator.forward` with use_norm=False (pix2pix_networks.py:604-631):
    Conv2d(k 4, padding 2, stride 2, bias) + LeakyReLU(0.1) for every entry of num_filters[:-1],
    then Conv2d(num_filters[-1] -> 1, k 4, stride 1, padding 2).  `sd` holds net.{0,2,4,...}.{weight,bias}."""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("net.")})
    for i in idx[:-1]:
        x = F.leaky_relu(F.conv2d(x, sd[f"net.{i}.weight"], sd[f"net.{i}.bias"], stride=2, padding=2), 0.1)
    i = idx[-1]
    return F.conv2d(x, sd[f"net.{i}.weight"], sd[f"net.{i}.bias"], stride=1, padding=2)


def gradient_loss(gen: torch.Tensor, gt: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """`Gradient_Loss` (losses_utils.py:30-61): backward differences with a zero on the left / top, SUMMED over
    the channels by the [-1, 1] filters (one output channel), |gt - gen| ** alpha, mean over [B,1,H,W]."""
    def dxy(t):
        s = t.sum(1, keepdim=True)
        dx = s - F.pad(s, (1, 0, 0, 0))[..., :, :-1]
        dy = s - F.pad(s, (0, 0, 1, 0))[..., :-1, :]
        return dx, dy
    gx, gy = dxy(gen)
    tx, ty = dxy(gt)
    return ((tx - gx).abs() ** alpha + (ty - gy).abs() ** alpha).mean()


def adversarial_loss(fake_outputs: torch.Tensor) -> torch.Tensor:
    """`Adversarial_Loss` (losses_utils.py:100-104), least-squares GAN"""
    return ((fake_outputs - 1) ** 2 / 2).mean()


def discriminate_loss(real_outputs: torch.Tensor, fake_outputs: torch.Tensor) -> torch.Tensor:
    """`Discriminate_Loss` (losses_utils.py:106-110)"""
    return ((real_outputs - 1) ** 2 / 2).mean() + (fake_outputs ** 2 / 2).mean()


def flow_loss(gen_flows: torch.Tensor, gt_flows: torch.Tensor) -> torch.Tensor:
    """`Flow_Loss` (losses_utils.py:10-15)"""
    return (gen_flows - gt_flows).abs().mean()


def generator_loss_full(out, rgb_t, op_t, d_gen, flow_pred=None, flow_gt=None, lam_adv=0.05, lam_gdl=1.0,
                        lam_flow=2.0, lam_lp=1.0, lam_lp_op=1.0, lam_latent=1.0) -> torch.Tensor:
    """`Twostream_vq_Loss.forward` (loss_zoo.py:310-336).  The FlowNet2-SD term (out of scope)
    enters only through precomputed flows; the latent term is the sum of the two commit scores (see
    `generator_loss`)."""
    rgb, op, (rd, od), _ = out[:4]
    loss = lam_adv * adversarial_loss(d_gen) + lam_gdl * gradient_loss(rgb, rgb_t) + \
        lam_lp * intensity_l2(rgb, rgb_t) + lam_lp_op * intensity_l2(op, op_t) + lam_latent * (rd + od).sum()
    if flow_pred is not None:
        loss = loss + lam_flow * flow_loss(flow_pred, flow_gt)
    return loss


def clone_state(sd: State, requires_grad: bool = False) -> State:
    out = {}
    for key, v in sd.items():
        c = v.detach().clone()
        leaf = key.rsplit(".", 1)[-1]
        is_buffer = leaf in ("running_mean", "running_var", "num_batches_tracked",
                             "embed", "cluster_size", "embed_avg")
        if requires_grad and not is_buffer and c.is_floating_point():
            c.requires_grad_(True)
        out[key] = c
    return out


# ---- FlowNet2-SD, the frozen flow estimator of the flow-consistency term -------------------------

def flownet2sd_forward(sd: State, inputs: torch.Tensor, rgb_max: float = 255.0, div_flow: float = 20.0) -> torch.Tensor:
    """`FlowNet2SD.forward` in eval mode (models/flownet2/models.py:15-59 over FlowNetSD.py:12-58, submodules.py:9-45,
    batchNorm=False): inputs [B,3,2,H,W] in 0..255 -> flow [B,2,H,W].  conv = Conv2d(k3, pad 1, bias) + LeakyReLU(0.1);
    deconv = ConvTranspose2d(k4, s2, p1, bias) + LeakyReLU(0.1); i_conv / predict_flow = bare Conv2d(k3);
    upsampled_flow = ConvTranspose2d(2, 2, 4, 2, 1)."""
    def conv(name, x, stride=1):
        return F.leaky_relu(F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=stride, padding=1), 0.1)

    def deconv(name, x):
        return F.leaky_relu(F.conv_transpose2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1), 0.1)

    def bare(name, x):
        return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=1)

    def up(name, x):
        return F.conv_transpose2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=2, padding=1)

    b = inputs.shape[0]
    rgb_mean = inputs.contiguous().view(b, 3, -1).mean(dim=-1).view(b, 3, 1, 1, 1)
    x = (inputs - rgb_mean) / rgb_max
    x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
    c0 = conv("conv0", x)
    c1 = conv("conv1_1", conv("conv1", c0, 2))
    c2 = conv("conv2_1", conv("conv2", c1, 2))
    c3 = conv("conv3_1", conv("conv3", c2, 2))
    c4 = conv("conv4_1", conv("conv4", c3, 2))
    c5 = conv("conv5_1", conv("conv5", c4, 2))
    c6 = conv("conv6_1", conv("conv6", c5, 2))
    flow6 = bare("predict_flow6", c6)
    cat5 = torch.cat((c5, deconv("deconv5", c6), up("upsampled_flow6_to_5", flow6)), 1)
    flow5 = bare("predict_flow5", F.conv2d(cat5, sd["inter_conv5.0.weight"], sd["inter_conv5.0.bias"], padding=1))
    cat4 = torch.cat((c4, deconv("deconv4", cat5), up("upsampled_flow5_to_4", flow5)), 1)
    flow4 = bare("predict_flow4", F.conv2d(cat4, sd["inter_conv4.0.weight"], sd["inter_conv4.0.bias"], padding=1))
    cat3 = torch.cat((c3, deconv("deconv3", cat4), up("upsampled_flow4_to_3", flow4)), 1)
    flow3 = bare("predict_flow3", F.conv2d(cat3, sd["inter_conv3.0.weight"], sd["inter_conv3.0.bias"], padding=1))
    cat2 = torch.cat((c2, deconv("deconv2", cat3), up("upsampled_flow3_to_2", flow3)), 1)
    flow2 = bare("predict_flow2", F.conv2d(cat2, sd["inter_conv2.0.weight"], sd["inter_conv2.0.bias"], padding=1))
    return F.interpolate(flow2 * div_flow, scale_factor=4, mode="bilinear", align_corners=False)
