"""bench.py's own control flow on the GPU box: the N > 1 launch (two ranks sharing GPU 0 over gloo, exactly as the driver
launches it apart from the backend) and the N = 1 launch, both on the 2-layer --tiny model, must print ONE JSON line with the
contract's fields and report the replayed-segments launch mode."""
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


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, stdout[-3000:]
    return json.loads(lines[0])


def test_two_rank_bench_flow_prints_one_json_line():
    env = dict(os.environ, X2_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--tiny",
           "--batch", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert out["launch_mode"] == "hipgraph-segments", out["launch_mode"]
    assert out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert "cpu_baseline" not in out                        # rank 0 at N = 1 only


def test_single_rank_bench_flow_tiny():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--tiny", "--batch", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["launch_mode"] == "hipgraph-segments" and out["roofline"]["frac"] > 0
    assert all(abs(v) < 1e4 for v in out["config"]["losses"].values())
