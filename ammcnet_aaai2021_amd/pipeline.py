"""Input pipeline for the path: decoded frames / `.flo` files -> device-resident, normalised
sub-videos, one upload per frame.

The reference's `test_dataset` (Code/dataset/two_stream_dataset.py:491-539) decodes, resizes and normalises every
frame once per clip that contains it (5x for rgb, 4x for flow) on DataLoader workers, and ships fp32 clips
(3.4 MB per clip) through `.cuda()`.  Here a frame crosses PCIe once, as the raw decoded bytes (uint8 RGB at the
native resolution; the `.flo` payload as stored), and the resize / ToTensor / Normalize arithmetic runs on the GPU
(`csrc/pipeline.hip`).  Clips are then slices of the resident `[T, c, 256, 256]` tensors (`harness.score_batch`).

`SubVideoStager` overlaps the next sub-video's host read + H2D copy (pinned staging buffers, a side HIP stream, one
event per sub-video) with the scoring of the current one.  One stager per process = one loader shard per GPU.

Host-side decoding: `.flo` is parsed here (numpy); JPEG/PNG frames go through PIL when it is installed (the reference
uses TurboJPEG, `utils/img_process.py:6-19`); `.npy` frame stacks need nothing.
"""
from __future__ import annotations

import glob
import os
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

FLO_MAGIC = np.float32(202021.25)


def read_flo(path: str) -> np.ndarray:
    """Middlebury .flo -> float32 [h, w, 2] (the reference's `readFlow`, utils/flowlib.py:589-611); raises on a bad
    magic number instead of returning None"""
    with open(path, "rb") as f:
        head = np.fromfile(f, np.float32, count=1)
        if head.size != 1 or head[0] != FLO_MAGIC:
            raise ValueError(f"{path}: magic number incorrect, not a .flo file")
        wh = np.fromfile(f, np.int32, count=2)
        if wh.size != 2 or wh[0] <= 0 or wh[1] <= 0:
            raise ValueError(f"{path}: bad .flo header")
        w, h = int(wh[0]), int(wh[1])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    if data.size != 2 * w * h:
        raise ValueError(f"{path}: truncated .flo payload")
    return data.reshape(h, w, 2)


def read_image(path: str) -> np.ndarray:
    """decoded RGB uint8 [h, w, 3]"""
    if path.endswith(".npy"):
        a = np.load(path)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"{path}: expected uint8 [h, w, 3]")
        return a
    try:
        from PIL import Image
    except ImportError as e:                                    # pragma: no cover
        raise RuntimeError("decoding image files needs PIL; store frames as uint8 .npy instead") from e
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def list_subvideos(rgb_root: str, op_root: str) -> List[Tuple[List[str], List[str]]]:
    """sorted sub-video folders, sorted files inside (test_helper.py:404-411, two_stream_dataset.py:515-518)"""
    out = []
    for name in sorted(os.listdir(rgb_root)):
        frames = sorted(glob.glob(os.path.join(rgb_root, name, "*")))
        flows = sorted(glob.glob(os.path.join(op_root, name, "*")))
        out.append((frames, flows))
    return out


def frames_to_device(frames_u8: torch.Tensor, size: Tuple[int, int] = (256, 256), bgr: bool = False) -> torch.Tensor:
    """uint8 [T, h, w, 3] on the GPU -> float32 [T, 3, H, W] in [-1, 1] (`_load_frame` + ToTensor + Normalize)"""
    if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3:
        raise _lib.AmmcHipError("frames_to_device: expected a uint8 [T, h, w, 3] GPU tensor (no CPU path)")
    frames_u8 = frames_u8.contiguous()
    t, h, w, _ = frames_u8.shape
    ow, oh = size
    out = torch.empty(t, 3, oh, ow, device=frames_u8.device, dtype=torch.float32)
    s = torch.cuda.current_stream(frames_u8.device).cuda_stream
    _lib.check(_lib.load().ammc_frames_u8_to_f32(frames_u8.data_ptr(), t, h, w, out.data_ptr(), oh, ow, int(bgr), s),
               "frames_u8_to_f32")
    return out


def flows_to_device(flows: torch.Tensor, size: Tuple[int, int] = (256, 256)) -> torch.Tensor:
    """float32 [T, h, w, 2] on the GPU -> float32 [T, 2, H, W] (`_load_op`)"""
    if not flows.is_cuda or flows.dtype != torch.float32 or flows.dim() != 4 or flows.shape[3] != 2:
        raise _lib.AmmcHipError("flows_to_device: expected a float32 [T, h, w, 2] GPU tensor (no CPU path)")
    flows = flows.contiguous()
    t, h, w, _ = flows.shape
    ow, oh = size
    out = torch.empty(t, 2, oh, ow, device=flows.device, dtype=torch.float32)
    s = torch.cuda.current_stream(flows.device).cuda_stream
    _lib.check(_lib.load().ammc_flows_to_f32(flows.data_ptr(), t, h, w, out.data_ptr(), oh, ow, s), "flows_to_f32")
    return out


def load_subvideo_host(frame_files: Sequence[str], flow_files: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """host side of one sub-video: decoded frames uint8 [T, h, w, 3], flows float32 [T', h, w, 2]"""
    frames = np.stack([read_image(p) for p in frame_files])
    flows = np.stack([read_flo(p) if p.endswith(".flo") else np.load(p).astype(np.float32) for p in flow_files])
    return frames, flows


class SubVideoStager:
    """Iterate device-resident (rgb [T,3,H,W], flow [T',2,H,W]) pairs; sub-video i+1 is read and copied to the GPU on a
    side stream while the caller works on sub-video i.

    `sources`: a sequence of callables returning (frames uint8 [T,h,w,3], flows float32 [T',h,w,2]) numpy arrays (or a
    list of (frame_files, flow_files) from `list_subvideos`); pinned uint8 / float32 tensors are uploaded from where they
    are.  `shard=(rank, world)` keeps every world-th sub-video; `ahead`: sub-videos staged beyond the current one.
    """

    def __init__(self, sources: Sequence, device, size: Tuple[int, int] = (256, 256), bgr: bool = False,
                 shard: Tuple[int, int] = (0, 1), ahead: int = 1, timed: bool = False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AmmcHipError("SubVideoStager stages onto a GPU; there is no CPU pipeline")
        rank, world = shard
        self.sources = [s for i, s in enumerate(sources) if i % world == rank]
        self.size, self.bgr = size, bgr
        self.ahead = max(1, int(ahead))        # sub-videos staged beyond the one being scored
        self.timed, self.stage_events = bool(timed), []
        self.stream = torch.cuda.Stream(self.device)
        self.bytes_uploaded = 0
        self.host_seconds = 0.0        # file reads / decoding / pinning in the reader thread (overlapped with the GPU)

    def _host(self, src):
        import time
        t0 = time.perf_counter()
        frames, flows = src() if callable(src) else load_subvideo_host(*src)
        pf, po = self._pinned(frames, torch.uint8), self._pinned(flows, torch.float32)
        self.host_seconds += time.perf_counter() - t0
        return pf, po

    @staticmethod
    def _pinned(a, dtype) -> torch.Tensor:
        """page-locked host tensor of `a` (numpy array or tensor); a source that already hands out pinned tensors of the
        right type - a decoder writing into its own staging buffers - is taken as it is"""
        if isinstance(a, torch.Tensor):
            if a.dtype == dtype and a.is_contiguous() and a.is_pinned():
                return a
            return a.to(dtype).contiguous().pin_memory()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8 if dtype == torch.uint8 else np.float32)).pin_memory()

    def _reader(self, q):
        """background thread: file reads, decoding and pinning, one sub-video ahead of the GPU staging"""
        try:
            for src in self.sources:
                q.put(self._host(src))
        except BaseException as e:                               # surfaced in the consumer
            q.put(e)

    def _stage(self, host):
        """pinned buffers -> async H2D + the device kernels, all on the side stream"""
        pf, po = host
        self.bytes_uploaded += pf.numel() + 4 * po.numel()
        with torch.cuda.stream(self.stream):
            if self.timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
            df = pf.to(self.device, non_blocking=True)
            do = po.to(self.device, non_blocking=True)
            if self.timed:
                e1.record(self.stream)
            rgb = frames_to_device(df, self.size, self.bgr)
            op = flows_to_device(do, self.size)
            ev = torch.cuda.Event(enable_timing=self.timed)
            ev.record(self.stream)
            if self.timed:
                self.stage_events.append((e0, e1, ev))
        return rgb, op, ev, (pf, po, df, do)

    def stage_ms(self) -> Tuple[float, float]:
        """(copy, kernels) milliseconds summed over the staged sub-videos (`timed=True`; call after a device sync)"""
        return (sum(a.elapsed_time(b) for a, b, _ in self.stage_events), sum(b.elapsed_time(c) for _, b, c in self.stage_events))

    def __len__(self) -> int:
        return len(self.sources)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        import collections
        import queue
        import threading
        if not self.sources:
            return
        q: "queue.Queue" = queue.Queue(maxsize=1 + self.ahead)
        th = threading.Thread(target=self._reader, args=(q,), daemon=True)
        th.start()

        def take():
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            return self._stage(item)

        # `ahead` sub-videos are staged (H2D + conversion kernels on the side stream) beyond the one handed out
        staged = collections.deque()
        taken = 0
        for i in range(len(self.sources)):
            while taken < len(self.sources) and len(staged) < 1 + self.ahead:
                staged.append(take())
                taken += 1
            rgb, op, ev, keep = staged.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                                    # the consumer's stream waits, not the host
            rgb.record_stream(cur)
            op.record_stream(cur)
            yield rgb, op
            del keep
        th.join()
