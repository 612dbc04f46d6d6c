"""Deterministic synthetic parameters and clips for the appearance-motion path.

There are no trained checkpoints or datasets for the reference
(/root/reference/.MISSING_LARGE_BLOBS), so every parity test and the benchmark
run on closed-form synthetic data.  Everything here is pure integer hashing
(splitmix64) mapped to floats, so the values are bit-identical on every
machine, numpy and torch version: golden fixtures made in the authoring
container stay valid on the GPU box.

Shapes follow the reference state_dict schema of `twostream`
(Code/models/unet.py:967-980; 222 entries) and the input conventions of the
eval loop (Code/run_helper/test_helper.py:428-438).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def hashed_uniform(tag: str, shape, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """U(lo, hi) fp32 tensor that depends only on (tag, shape); 24-bit grid."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(tag.encode()) * 2654435761 % (1 << 63))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + seed * np.uint64(0x100000001B3)
        bits = _splitmix64(idx) >> np.uint64(40)          # 24 random bits
    u = bits.astype(np.float64) / float(1 << 24)           # exact in f64
    v = (lo + (hi - lo) * u).astype(np.float32)
    return torch.from_numpy(v.reshape(tuple(shape)))


def hashed_normal(tag: str, shape, std: float = 1.0) -> torch.Tensor:
    """Approximately normal (Irwin-Hall of 4 uniforms), exact-integer based."""
    acc = None
    for j in range(4):
        u = hashed_uniform(f"{tag}#n{j}", shape, -1.0, 1.0).double()
        acc = u if acc is None else acc + u
    # var of sum of 4 U(-1,1) = 4/3
    return (acc * (std / (4.0 / 3.0) ** 0.5)).float()


# ----------------------------------------------------------------------------
# state_dict schema
# ----------------------------------------------------------------------------

def _double_conv_entries(prefix: str, cin: int, cout: int):
    yield f"{prefix}.0.weight", (cout, cin, 3, 3)
    yield from _bn_entries(f"{prefix}.1", cout)
    yield f"{prefix}.3.weight", (cout, cout, 3, 3)
    yield from _bn_entries(f"{prefix}.4", cout)


def _bn_entries(prefix: str, c: int):
    yield f"{prefix}.weight", (c,)
    yield f"{prefix}.bias", (c,)
    yield f"{prefix}.running_mean", (c,)
    yield f"{prefix}.running_var", (c,)
    yield f"{prefix}.num_batches_tracked", ()


def unet_schema(prefix: str, cin: int, cout: int, embed_dim=None, n_embed=None, k=None):
    """(key, shape) pairs of `UNet` / `UNetMem_v7` (unet.py:61-83, 908-937)."""
    p = prefix
    yield from _double_conv_entries(f"{p}inc.conv.conv", cin, 64)
    yield from _double_conv_entries(f"{p}down1.mpconv.1.conv", 64, 128)
    yield from _double_conv_entries(f"{p}down2.mpconv.1.conv", 128, 256)
    yield from _double_conv_entries(f"{p}down3.mpconv.1.conv", 256, 512)
    for name, c_in, c_out in (("up1", 512, 256), ("up2", 256, 128), ("up3", 128, 64)):
        yield f"{p}{name}.up.weight", (c_in, c_in // 2, 2, 2)
        yield f"{p}{name}.up.bias", (c_in // 2,)
        yield from _double_conv_entries(f"{p}{name}.conv.conv", c_in, c_out)
    yield f"{p}outc.weight", (cout, 64, 3, 3)
    yield f"{p}outc.bias", (cout,)
    if embed_dim is not None:
        q = f"{p}vq_down3.quan"
        yield f"{q}.enc.weight", (embed_dim, 512, 1, 1)
        yield f"{q}.enc.bias", (embed_dim,)
        yield f"{q}.quantize.embed", (embed_dim, n_embed)
        yield f"{q}.quantize.cluster_size", (n_embed,)
        yield f"{q}.quantize.embed_avg", (embed_dim, n_embed)
        yield f"{q}.dec.weight", (512, embed_dim * k, 1, 1)
        yield f"{q}.dec.bias", (512,)


def twostream_schema(in_channel=(12, 6), out_channel=(3, 2), embed_dim=64, n_embed=256, k=2):
    """(key, shape) pairs of `twostream` (unet.py:967-980), reference order."""
    yield from unet_schema("rgb.", in_channel[0], out_channel[0], embed_dim, n_embed, k)
    yield from unet_schema("op.", in_channel[1], out_channel[1], embed_dim, n_embed, k)
    yield from _double_conv_entries("bridge.O2F.conv", 512, 512)
    yield from _double_conv_entries("bridge.F20.conv", 512, 512)


def _fill(key: str, shape, tag: str) -> torch.Tensor:
    leaf = key.rsplit(".", 1)[-1]
    t = f"{tag}:{key}"
    if leaf == "num_batches_tracked":
        return torch.tensor(0, dtype=torch.int64)
    if leaf == "running_mean":
        return hashed_uniform(t, shape, -0.2, 0.2)
    if leaf == "running_var":
        return hashed_uniform(t, shape, 0.6, 1.4)
    if leaf == "cluster_size":
        return hashed_uniform(t, shape, 0.5, 4.0)
    if leaf == "embed":
        return hashed_normal(t, shape, 0.9)
    if leaf == "embed_avg":
        # consistent with embed = embed_avg / cluster_size is NOT required by
        # the reference (unet.py:277-280 starts them equal); keep them close.
        return hashed_normal(t.replace("embed_avg", "embed"), shape, 0.9) * \
            hashed_uniform(t, (1, shape[1]), 0.5, 4.0)
    if len(shape) == 1:
        parent = key.rsplit(".", 2)[-2]
        is_bn = parent in ("1", "4")
        if leaf == "weight" and is_bn:
            return hashed_uniform(t, shape, 0.7, 1.3)
        if leaf == "bias" and is_bn:
            return hashed_uniform(t, shape, -0.2, 0.2)
        return hashed_uniform(t, shape, -0.1, 0.1)          # conv / convT bias
    # conv weights: variance-preserving uniform (He), so that 20 layers deep the
    # activations stay O(1) and every layer matters to the output.
    if key.endswith("up.weight"):                           # ConvT [Cin, Cout, 2, 2]
        fan_in = shape[0]
        bound = (3.0 / fan_in) ** 0.5
    else:
        fan_in = shape[1] * shape[2] * shape[3]
        bound = (6.0 / fan_in) ** 0.5
        if ".enc." in key or ".dec." in key:
            bound = (3.0 / fan_in) ** 0.5
        if "outc" in key:                                   # keep tanh out of saturation
            bound = (3.0 / fan_in) ** 0.5 / 3.0
    return hashed_uniform(t, shape, -bound, bound)


def make_twostream_state(in_channel=(12, 6), out_channel=(3, 2), embed_dim=64, n_embed=256,
                         k=2, tag="ammc") -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for key, shape in twostream_schema(in_channel, out_channel, embed_dim, n_embed, k):
        sd[key] = _fill(key, shape, tag)
    return sd


def _fill_from_scratch(key: str, shape, tag: str, sd) -> torch.Tensor:
    leaf = key.rsplit(".", 1)[-1]
    t = f"{tag}:{key}"
    if leaf == "num_batches_tracked":
        return torch.tensor(0, dtype=torch.int64)
    if leaf == "running_mean":
        return torch.zeros(shape)
    if leaf == "running_var":
        return torch.ones(shape)
    if leaf == "cluster_size":
        return torch.zeros(shape)
    if leaf == "embed":
        return hashed_normal(t, shape, 1.0)
    if leaf == "embed_avg":
        return sd[key.replace("embed_avg", "embed")].clone()
    if len(shape) == 1:
        parent = key.rsplit(".", 2)[-2]
        if parent in ("1", "4"):                             # BatchNorm2d: gamma ~ N(1, 0.02), beta = 0
            return 1.0 + hashed_normal(t, shape, 0.02) if leaf == "weight" else torch.zeros(shape)
        wshape = sd[key[:-4] + "weight"].shape               # conv bias: torch's default U(-1/sqrt(fan_in), +), untouched by the init
        fan_in = wshape[1] * wshape[2] * wshape[3]           # (ConvTranspose2d [Cin, Cout, 2, 2]: torch takes dim 1 here as well)
        b = 1.0 / fan_in ** 0.5
        return hashed_uniform(t, shape, -b, b)
    return hashed_normal(t, shape, 0.02)                     # Conv2d / ConvTranspose2d weights ~ N(0, 0.02)


def make_from_scratch_state(in_channel=(12, 6), out_channel=(3, 2), embed_dim=64, n_embed=256, k=2,
                            tag="ammc-scratch") -> "OrderedDict[str, torch.Tensor]":
    """The state the reference's training starts from when no checkpoint is given, with its
    random draws replaced by the hash filler of the same distributions: `generator.apply(weights_init_normal)`
    (utils/utils.py:328-334, 342: Conv* weights ~ N(0, 0.02), BatchNorm2d gamma ~ N(1, 0.02), beta = 0) over torch's
    default construction (conv biases U(+-1/sqrt(fan_in)), running statistics 0 / 1) and `Quantize_topk.__init__`
    (models/unet.py:277-280: embed ~ N(0, 1), **cluster_size = 0, embed_avg = embed**).  tests/golden/make_golden.py
    (`from_scratch`) checks structure and distribution parameters against a model the reference initialised itself."""
    sd = OrderedDict()
    for key, shape in twostream_schema(in_channel, out_channel, embed_dim, n_embed, k):
        sd[key] = _fill_from_scratch(key, shape, tag, sd)
    return sd


def make_unet_state(cin=12, cout=3, tag="ammc-unet") -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for key, shape in unet_schema("", cin, cout):
        sd[key] = _fill(key, shape, tag)
    return sd


def discriminator_schema(input_nc=3, num_filters=(128, 256, 512, 512)):
    """state_dict entries of `PixelDiscriminator(input_nc, num_filters, use_norm=False)`
    (reference pix2pix_networks.py:604-631): convs at net.0, net.2, ..., LeakyReLU in between"""
    chans = [input_nc] + list(num_filters[:-1])
    for i in range(len(chans) - 1):
        yield f"net.{2 * i}.weight", (chans[i + 1], chans[i], 4, 4)
        yield f"net.{2 * i}.bias", (chans[i + 1],)
    last = 2 * (len(chans) - 1)
    yield f"net.{last}.weight", (1, num_filters[-1], 4, 4)
    yield f"net.{last}.bias", (1,)


def make_discriminator_state(input_nc=3, num_filters=(128, 256, 512, 512), tag="ammc-d"):
    sd = OrderedDict()
    for key, shape in discriminator_schema(input_nc, num_filters):
        t = f"{tag}:{key}"
        if len(shape) == 1:
            sd[key] = hashed_uniform(t, shape, -0.1, 0.1)
        else:
            bound = (6.0 / (shape[1] * 16)) ** 0.5
            sd[key] = hashed_uniform(t, shape, -bound, bound)
    return sd


def flownet2sd_schema():
    """state_dict entries of `FlowNet2SD(batchNorm=False)` (reference models/flownet2/FlowNetSD.py:12-45; 45,371,666
    parameters): Conv2d k3 with bias inside Sequentials (`.0.`), ConvTranspose2d k4 [in, out, 4, 4]"""
    convs = [("conv0", 6, 64), ("conv1", 64, 64), ("conv1_1", 64, 128), ("conv2", 128, 128), ("conv2_1", 128, 128),
             ("conv3", 128, 256), ("conv3_1", 256, 256), ("conv4", 256, 512), ("conv4_1", 512, 512),
             ("conv5", 512, 512), ("conv5_1", 512, 512), ("conv6", 512, 1024), ("conv6_1", 1024, 1024)]
    for name, ci, co in convs:
        yield f"{name}.0.weight", (co, ci, 3, 3)
        yield f"{name}.0.bias", (co,)
    for name, ci, co in (("deconv5", 1024, 512), ("deconv4", 1026, 256), ("deconv3", 770, 128), ("deconv2", 386, 64)):
        yield f"{name}.0.weight", (ci, co, 4, 4)
        yield f"{name}.0.bias", (co,)
    for name, ci, co in (("inter_conv5", 1026, 512), ("inter_conv4", 770, 256), ("inter_conv3", 386, 128),
                         ("inter_conv2", 194, 64)):
        yield f"{name}.0.weight", (co, ci, 3, 3)
        yield f"{name}.0.bias", (co,)
    for lvl, ci in ((6, 1024), (5, 512), (4, 256), (3, 128), (2, 64)):
        yield f"predict_flow{lvl}.weight", (2, ci, 3, 3)
        yield f"predict_flow{lvl}.bias", (2,)
    for name in ("upsampled_flow6_to_5", "upsampled_flow5_to_4", "upsampled_flow4_to_3", "upsampled_flow3_to_2"):
        yield f"{name}.weight", (2, 2, 4, 4)
        yield f"{name}.bias", (2,)


def make_flownet2sd_state(tag="ammc-flow"):
    """variance-preserving uniform weights (the published checkpoint is not available), small biases"""
    sd = OrderedDict()
    for key, shape in flownet2sd_schema():
        t = f"{tag}:{key}"
        if len(shape) == 1:
            sd[key] = hashed_uniform(t, shape, -0.1, 0.1)
        else:
            transposed = key.startswith(("deconv", "upsampled"))
            fan_in = (shape[0] * 4 if transposed else shape[1] * 9)          # k4 s2: 4 taps reach an output pixel
            bound = (6.0 / (1.01 * fan_in)) ** 0.5                            # LeakyReLU(0.1) gain
            sd[key] = hashed_uniform(t, shape, -bound, bound)
    return sd


# ----------------------------------------------------------------------------
# synthetic clips (reference loader two_stream_dataset.py:72-99)
# ----------------------------------------------------------------------------

def make_clips(batch: int, height: int = 256, width: int = 256, in_channel=(12, 6),
               out_channel=(3, 2), tag="clip"):
    """rgb_x, op_x, rgb_target, op_target with the reference's value ranges.

    rgb is U(-1,1) (after Normalize(0.5,0.5)); a flow frame has channel 0 = u/256
    with u ~ N(0, 2px) and channel 1 = channel0/256, as `_load_op` produces.
    """
    rgb_x = hashed_uniform(f"{tag}:rgb", (batch, in_channel[0], height, width))
    rgb_t = hashed_uniform(f"{tag}:rgb_t", (batch, out_channel[0], height, width))
    nf = in_channel[1] // 2
    u = hashed_normal(f"{tag}:op", (batch, nf, 1, height, width), 2.0) / 256.0
    op_x = torch.cat([u, u / 256.0], dim=2).reshape(batch, in_channel[1], height, width)
    ut = hashed_normal(f"{tag}:op_t", (batch, 1, height, width), 2.0) / 256.0
    op_t = torch.cat([ut, ut / 256.0], dim=1)[:, : out_channel[1]]
    return rgb_x.contiguous(), op_x.contiguous(), rgb_t.contiguous(), op_t.contiguous()
