"""Two ranks against the REAL library (pytest -m gpu): two child processes share GPU 0, exchange over gloo, and run the
three shard modes of the maximiser, the tracked sequential batch, the gradient maximiser and the sharded fitters through
libbosship.so (tests/two_rank_worker.py); plus bench.py's own N>1 launcher (`python bench.py --gpus 2`)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_through_the_library():
    import __graft_entry__ as entry
    entry.build()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py")], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=500)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r}: ok" in out, f"rank {r} failed:\n{out[-4000:]}"


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun): the launcher spawns the ranks itself; on a 1-GPU box they share the
    device and exchange over gloo (`rehearsal`).  The line carries a weak and a strong record."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["acq_evals_per_sec"] > 0
    s = out["strong_scaling"]
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["M_per_gpu"] == 4096 and s["value"] > 0
    assert out["roofline"]["frac"] > 0.3
