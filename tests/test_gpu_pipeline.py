"""Device half of the input pipeline (csrc/pipeline.hip) against oracle/pipeline_oracle.py: bit-exact (the kernels
follow the oracle's arithmetic operation by operation), and the double-buffered stager end to end."""
import numpy as np
import pytest
import torch

from ammcnet_aaai2021_amd import harness as Hn, pipeline as P, synthetic as S
from oracle import pipeline_oracle as PO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("h,w,oh,ow", [(240, 360, 256, 256), (360, 640, 256, 256), (256, 256, 256, 256),
                                        (158, 238, 256, 256), (480, 856, 64, 96), (3, 5, 8, 8)])
def test_frames_and_flows_bit_exact(h, w, oh, ow):
    rng = np.random.default_rng(h * 1000 + w)
    frames = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    flows = rng.normal(0, 3, (2, h, w, 2)).astype(np.float32)
    got = P.frames_to_device(torch.from_numpy(frames).to(DEV), (ow, oh)).cpu().numpy()
    want = np.stack([PO.load_frame(f, (ow, oh)) for f in frames])
    assert got.shape == want.shape == (3, 3, oh, ow)
    assert np.array_equal(got, want)
    got_bgr = P.frames_to_device(torch.from_numpy(frames[..., ::-1].copy()).to(DEV), (ow, oh), bgr=True).cpu().numpy()
    assert np.array_equal(got_bgr, want)
    gf = P.flows_to_device(torch.from_numpy(flows).to(DEV), (ow, oh)).cpu().numpy()
    wf = np.stack([PO.load_op(f.copy(), (ow, oh)) for f in flows])
    assert np.array_equal(gf, wf)


def test_stager_order_sharding_and_scoring():
    rng = np.random.default_rng(11)
    vids = []
    for i in range(5):
        t = 9 + i
        vids.append((rng.integers(0, 256, (t, 120, 160, 3), dtype=np.uint8),
                     rng.normal(0, 2, (t - 1, 120, 160, 2)).astype(np.float32)))
    sources = [(lambda v=v: v) for v in vids]
    st = P.SubVideoStager(sources, DEV, size=(64, 64))
    seen = list(st)
    assert len(seen) == 5 and st.bytes_uploaded == sum(f.size + 4 * o.size for f, o in vids)
    for (rgb, op), (f, o) in zip(seen, vids):
        assert np.array_equal(rgb.cpu().numpy(), np.stack([PO.load_frame(x, (64, 64)) for x in f]))
        assert np.array_equal(op.cpu().numpy(), np.stack([PO.load_op(x.copy(), (64, 64)) for x in o]))
    part = list(P.SubVideoStager(sources, DEV, size=(64, 64), shard=(1, 2)))
    assert len(part) == 2 and torch.equal(part[0][0], seen[1][0]) and torch.equal(part[1][0], seen[3][0])

    def boom():
        raise IOError("unreadable sub-video")
    with pytest.raises(IOError):
        list(P.SubVideoStager([sources[0], boom], DEV, size=(64, 64)))
    # the staged tensors feed the evaluation loop directly
    import ammcnet_aaai2021_amd as A
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(S.make_twostream_state())
    net = net.to(DEV).eval()
    rec = Hn.evaluate_subvideo(net, seen[0][0], seen[0][1])
    assert len(rec["rgb_psnr"]) == vids[0][0].shape[0] and np.isfinite(rec["rgb_psnr"]).all()


def test_raw_frames_to_records_equal_the_oracle_loop():
    """The evaluation loop end to end as `bench.py`'s eval_e2e leg runs it - pinned uint8 frames + flow payloads ->
    SubVideoStager (two ahead) -> evaluate_stream (clips as overlapping windows read in place by the first-layer
    kernel, one score copy per sub-video) - against the oracle: loaders (oracle/pipeline_oracle.py) -> model forward
    (oracle/ammc_oracle.py) -> the restated loop of run_helper/test_helper.py:408-473.  Records to 1e-4 (PSNR is a
    log of a mean of squares of 1e-4-accurate frames; commit values are means over the batch)."""
    import ammcnet_aaai2021_amd as A
    from oracle import ammc_oracle as O
    rng = np.random.default_rng(5)
    lens = (22, 9)                                               # 18 clips = 16 + 2, and 5 clips (one short batch)
    vids = []
    for t in lens:
        fr = torch.from_numpy(rng.integers(0, 256, (t, 60, 90, 3), dtype=np.uint8)).pin_memory()
        fl = torch.from_numpy(rng.normal(0, 2, (t - 1, 60, 90, 2)).astype(np.float32)).pin_memory()
        vids.append((fr, fl))
    sd = S.make_twostream_state()
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    info = {}
    st = P.SubVideoStager([(lambda v=v: v) for v in vids], DEV, size=(64, 64), ahead=2, timed=True)
    got = Hn.evaluate_stream(net, st, "ped2", stats=info)
    torch.cuda.synchronize()
    assert info == {"score_copies": 2, "rerun_batches": 0} and len(st.stage_events) == 2 and min(st.stage_ms()) > 0
    assert all(s.first_mid is not None for s in net._engine._last["streams"])      # the windows were read in place
    osd = O.clone_state(sd)
    for v, (fr, fl) in enumerate(vids):
        rgb = torch.from_numpy(np.stack([PO.load_frame(x, (64, 64)) for x in fr.numpy()]))
        op = torch.from_numpy(np.stack([PO.load_op(x.copy(), (64, 64)) for x in fl.numpy()]))
        want = O.eval_subvideo_records(lambda a, b: O.twostream_forward(osd, a, b, 2), rgb, op)
        for key, name in (("rgb_img_pred_records", "rgb_psnr"), ("rgb_fea_comm_records", "rgb_comm"),
                          ("op_img_pred_records", "op_psnr"), ("op_fea_comm_records", "op_comm")):
            assert np.allclose(got[key][v], want[name], rtol=1e-4, atol=0), (v, key)
    # the same records when the sub-videos are resident and scored through the batch-sharded loop
    res = list(P.SubVideoStager([(lambda v=v: v) for v in vids], DEV, size=(64, 64)))
    full = Hn.evaluate_dataset(net, res, "ped2")
    # (commit values bit for bit; the PSNR sums are accumulated with one fp32 atomic per output tile, whose order is free)
    for key in ("rgb_fea_comm_records", "op_fea_comm_records"):
        for a, b in zip(got[key], full[key]):
            assert np.array_equal(a, b), key
    for key in ("rgb_img_pred_records", "op_img_pred_records"):
        for a, b in zip(got[key], full[key]):
            assert np.allclose(a, b, rtol=1e-5, atol=0), key
