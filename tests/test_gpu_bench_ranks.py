"""bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), on the one GPU of the
test box: both ranks share cuda:0 and talk over gloo (AMMC_BENCH_SHARE_GPU=1).  Checks the single JSON line, the
whole-job aggregation and both modes."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,extra", [("infer", ["--batch", "4", "--size", "64", "--no-cpu-baseline"]),
                                         ("train", ["--batch", "2", "--size", "64"])])
def test_bench_two_ranks(mode, extra):
    env = dict(os.environ, AMMC_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--mode", mode] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    per_step = 2 * (4 if mode == "infer" else 2)                    # whole-job units per step = sum over ranks
    assert abs(rec["value"] - per_step / (rec["ms_per_step"] * 1e-3)) <= 0.02 * rec["value"]
    if mode == "infer":
        assert rec["roofline"]["bound"] == "mfma" and rec["roofline"]["frac"] > 0
