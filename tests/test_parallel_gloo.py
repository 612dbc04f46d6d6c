"""The N>1 HOST LOGIC on CPU: world_size 2, gloo.

What this tier proves: the gradient reducer's mechanics (bucketing, async all-reduce, in-place averaging, stage-fed
buckets) on plain tensors and a toy torch module, state broadcast, the partition of whole batches over ranks, the
statistics sync, and the scoping of the finite-vote collective to the data-parallel group.

What it does NOT prove: a data-parallel step of the PRODUCT model - that needs the HIP library (there is no CPU path) and
is covered on the GPU by tests/test_gpu_ddp.py (2 ranks sharing one GPU == the oracle on the whole batch; the RCCL
world-of-one runs the real collectives) and tests/test_gpu_bench_ranks.py (bench.py's self-spawned ranks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ammcnet_aaai2021_amd import parallel as P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, dev = P.init_distributed("gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    # --- reducer: tensors of odd sizes, several buckets, values depend on the rank
    torch.manual_seed(0)
    shapes = [(64, 12, 3, 3), (64,), (128, 64, 3, 3), (3,), (512, 128, 1, 1), (2, 2)]
    grads = [torch.full(s, float(rank + 1)) + torch.arange(int(torch.tensor(s).prod())).reshape(s) * 1e-3
             for s in shapes]
    want = [torch.full(s, 1.5) + torch.arange(int(torch.tensor(s).prod())).reshape(s) * 1e-3 for s in shapes]
    red = P.BucketedGradReducer(bucket_mb=0.1)
    red.push(grads[:3])
    red.push(grads[3:])
    red.finish()
    ok = all(torch.allclose(g, w_) for g, w_ in zip(grads, want)) and red.buckets_launched >= 2
    # --- broadcast_state: rank 1 starts different, ends equal to rank 0
    lin = torch.nn.BatchNorm2d(4)
    with torch.no_grad():
        lin.weight.fill_(rank + 1.0)
        lin.running_mean.fill_(rank + 3.0)
    P.broadcast_state(lin, 0)
    ok = ok and float(lin.weight[0]) == 1.0 and float(lin.running_mean[0]) == 3.0
    # --- inference sharding: whole batches, each owned once; records gathered everywhere
    mine = P.shard_batches(7, rank, world)
    rec = P.gather_records({i: [i * 10, rank] for i in mine}, world)
    ok = ok and sorted(rec) == list(range(7)) and all(rec[i][0] == i * 10 for i in rec)
    # --- a data-parallel step on a tiny torch model equals the single-process large-batch step
    torch.manual_seed(1)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(4, 2, 1))
    P.broadcast_state(model, 0)
    x = torch.arange(2 * world * 3 * 4 * 4, dtype=torch.float32).reshape(2 * world, 3, 4, 4).sin()
    xs = x[rank * 2:(rank + 1) * 2]
    model(xs).square().mean().backward()
    red = P.BucketedGradReducer(bucket_mb=25)
    red.push([p.grad for p in model.parameters()])
    red.finish()
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(4, 2, 1))
    ref.load_state_dict(model.state_dict())
    ref(x).square().mean().backward()
    ok = ok and all(torch.allclose(p.grad, q_.grad, atol=1e-6) for p, q_ in zip(model.parameters(), ref.parameters()))
    # --- the finite-loss verdict is collective: rank 1's loss is inf, BOTH ranks refuse the step (harness._FiniteWatch)
    from ammcnet_aaai2021_amd import harness as H
    w_ = torch.nn.Parameter(torch.ones(2))
    opt = torch.optim.SGD([w_], lr=1.0)
    w_.grad = torch.ones(2)
    watch = H._FiniteWatch(torch.tensor(float("inf") if rank == 1 else 1.0))
    try:
        watch.step(opt)
        refused = False
    except FloatingPointError:
        refused = True
    ok = ok and refused and float(w_[0]) == 1.0
    watch = H._FiniteWatch(torch.tensor(2.0), torch.tensor(3.0))       # all finite everywhere: the step is taken
    watch.step(opt)
    ok = ok and float(w_[0]) == 0.0
    # --- ... but only among ranks that share gradients or statistics (harness._watch_group): a model without a reducer
    # trains on its own inside the initialised world - no collective, each rank decides alone (a vote on the default
    # group would hang as soon as one rank does not train); with a reducer on a sub-group the vote runs in THAT group
    lone = torch.nn.Linear(2, 2)
    vote, group = H._watch_group(lone)
    ok = ok and vote is False and group is None
    w2 = torch.nn.Parameter(torch.ones(2))
    opt2 = torch.optim.SGD([w2], lr=1.0)
    w2.grad = torch.ones(2)
    watch = H._FiniteWatch(torch.tensor(float("inf") if rank == 1 else 1.0), group=group, vote=vote)
    try:
        watch.step(opt2)
        refused = False
    except FloatingPointError:
        refused = True
    ok = ok and refused == (rank == 1) and float(w2[0]) == (1.0 if rank == 1 else 0.0)
    subs = [dist.new_group([r]) for r in range(world)]                     # (every rank creates every group)
    P.attach_reducer(lone, P.BucketedGradReducer(group=subs[rank]))
    vote, group = H._watch_group(lone)
    ok = ok and vote is True and group is subs[rank]
    w2.grad = torch.ones(2)
    watch = H._FiniteWatch(torch.tensor(float("inf") if rank == 0 else 1.0), group=group, vote=vote)
    try:
        watch.step(opt2)
        refused = False
    except FloatingPointError:
        refused = True
    ok = ok and refused == (rank == 0)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_shard_batches_partition():
    for n in (0, 1, 5, 16, 123):
        for world in (1, 2, 3, 8):
            parts = [P.shard_batches(n, r, world) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_reducer_is_a_no_op():
    g = [torch.ones(3)]
    red = P.BucketedGradReducer()
    red.push(g)
    red.finish()
    assert torch.equal(g[0], torch.ones(3)) and red.buckets_launched == 0


def test_forced_reducer_runs_its_collectives_in_a_world_of_one():
    """`force=True` takes the real path (flat bucket, async all-reduce, scatter back) with one rank: what the RCCL
    world-size-1 GPU test relies on"""
    port = _free_port()
    dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}")
    try:
        g = [torch.arange(5.0), torch.ones(2, 3)]
        red = P.BucketedGradReducer(bucket_mb=1e-5, force=True)
        red.push(g)
        red.finish()
        assert red.buckets_launched == 2
        assert torch.equal(g[0], torch.arange(5.0)) and torch.equal(g[1], torch.ones(2, 3))
    finally:
        dist.destroy_process_group()
