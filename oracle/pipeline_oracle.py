"""CPU restatement (numpy) of the reference's input pipeline - TEST INFRASTRUCTURE ONLY, never imported by the product.

What it follows: `Code/dataset/two_stream_dataset.py:72-99` (`_load_frame`, `_load_op`), `:491-539` (`test_dataset`:
sliding clips, ToTensor + Normalize(0.5, 0.5)), `Code/utils/flowlib.py:589-611` (`readFlow`).  The resize inside
those loaders is `cv2.resize(img, (256, 256))` = INTER_LINEAR of OpenCV 4.1.1 (environment.yaml), a third-party
dependency that is NOT installed in this image.  Its published algorithm (modules/imgproc/src/resize.cpp, the generic
path) is restated here:

  * coordinates: fx = (dx + 0.5) * (src_w / dst_w) - 0.5 evaluated in double and rounded to float, sx = floor(fx),
    fx -= sx; sx < 0 -> (0, 0); sx >= src_w - 1 -> (src_w - 1, 0).  Same per row.
  * 8-bit: the two weights become shorts, round(w * 2048) (`INTER_RESIZE_COEF_SCALE`); the horizontal pass keeps
    ints S = p0*a0 + p1*a1; the vertical pass is ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
  * float: D = p0*a0 + p1*a1 horizontally, then S0*b0 + S1*b1, each product and sum rounded to float.

Parity status of the resize (round 5): cv2 cannot be obtained here (no network, no wheel in the image), so `cv2.resize`
itself has never been run against this file.  PINNED by an independent witness (tests/test_pipeline_host.py::
test_oracle_resize_against_independent_witnesses - torch's bilinear `interpolate`, align_corners=False, the same
published half-pixel definition): the sampling geometry of both paths (coordinates, taps, border clamp), the float
path to float rounding, the 8-bit path to less than one grey level everywhere.  PARITY UNPINNED, still: the 8-bit
kernel's own fixed-point rounding (the 2048-step weights and the `>> 4`, `>> 16`, `+ 2 >> 2` steps restated above,
which decide the ~12 % of pixels that are one level off the rounded exact value), and whether a given OpenCV build
takes its generic path at all (IPP / OpenCL builds differ from it by a grey level).  Everything else in this file is
pinned by the reference's code directly (plain arithmetic).
"""
from __future__ import annotations

import numpy as np

FLO_MAGIC = np.float32(202021.25)


def read_flo(path: str) -> np.ndarray:
    """Middlebury .flo -> float32 [h, w, 2]  (flowlib.py:589-611)"""
    with open(path, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != FLO_MAGIC:
            raise ValueError(f"{path}: not a .flo file")
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    return np.resize(data, (h, w, 2))


def write_flo(path: str, flow: np.ndarray) -> None:
    h, w, _ = flow.shape
    with open(path, "wb") as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        np.ascontiguousarray(flow, np.float32).tofile(f)


def _coords(src: int, dst: int):
    """per destination index: source index and the weight of its right/lower neighbour (float32)"""
    scale = src / dst
    idx = np.empty(dst, np.int32)
    frac = np.empty(dst, np.float32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0)
        idx[d], frac[d] = s, f
    return idx, frac


def resize_linear_u8(img: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """uint8 [h, w, c] -> uint8 [oh, ow, c], OpenCV generic INTER_LINEAR for 8-bit images"""
    h, w, c = img.shape
    sx, fx = _coords(w, ow)
    sy, fy = _coords(h, oh)
    ax1 = np.rint(fx.astype(np.float32) * np.float32(2048)).astype(np.int32)
    ax0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int32)
    by1 = np.rint(fy.astype(np.float32) * np.float32(2048)).astype(np.int32)
    by0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int32)
    src = img.astype(np.int32)
    x1 = np.minimum(sx + 1, w - 1)
    rows = src[:, sx, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]          # [h, ow, c] ints
    y1 = np.minimum(sy + 1, h - 1)
    s0, s1 = rows[sy], rows[y1]                                                               # [oh, ow, c]
    out = (((by0[:, None, None] * (s0 >> 4)) >> 16) + ((by1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_linear_f32(img: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """float32 [h, w, c] -> float32 [oh, ow, c], OpenCV generic INTER_LINEAR for float images"""
    h, w, c = img.shape
    sx, fx = _coords(w, ow)
    sy, fy = _coords(h, oh)
    a1, a0 = fx, (np.float32(1) - fx).astype(np.float32)
    b1, b0 = fy, (np.float32(1) - fy).astype(np.float32)
    src = img.astype(np.float32)
    x1 = np.minimum(sx + 1, w - 1)
    rows = (src[:, sx, :] * a0[None, :, None]).astype(np.float32) + (src[:, x1, :] * a1[None, :, None]).astype(np.float32)
    rows = rows.astype(np.float32)
    y1 = np.minimum(sy + 1, h - 1)
    out = (rows[sy] * b0[:, None, None]).astype(np.float32) + (rows[y1] * b1[:, None, None]).astype(np.float32)
    return out.astype(np.float32)


def load_frame(rgb_u8: np.ndarray, size=(256, 256)) -> np.ndarray:
    """decoded RGB frame uint8 [h, w, 3] -> float32 [3, H, W] in [-1, 1]: `_load_frame` + ToTensor + Normalize(0.5, 0.5)
    (two_stream_dataset.py:72-83, 503-506)"""
    img = resize_linear_u8(rgb_u8, size[1], size[0])
    t = img.astype(np.float32) / np.float32(255)
    return ((t - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1).copy()


def load_op(flow: np.ndarray, size=(256, 256)) -> np.ndarray:
    """.flo contents float32 [h, w, 2] -> float32 [2, H, W]: `_load_op` (two_stream_dataset.py:85-99), including its
    second channel being derived from the already scaled first one"""
    image_width, image_height = size
    img = resize_linear_f32(flow, image_height, image_width)
    img[:, :, 0] = img[:, :, 0] * np.float32(1.0) / np.float32(image_height)
    img[:, :, 1] = img[:, :, 0] * np.float32(1.0) / np.float32(image_width)
    return img.transpose(2, 0, 1).copy()


def clips(frames: np.ndarray, clip_length: int) -> np.ndarray:
    """[T, c, H, W] -> [T - clip_length + 1, clip_length, c, H, W]: `test_dataset.__getitem__` for every index"""
    n = frames.shape[0] - clip_length + 1
    return np.stack([frames[i:i + clip_length] for i in range(n)])
