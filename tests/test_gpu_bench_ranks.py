"""bench.py's N > 1 code path on a one-GPU box: the ranks share GPU 0 and the collective runs over gloo
(MOSAIC_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device).  The timings of such a run are void; what
is checked is that the launch contract holds: torch.distributed.run with N ranks, ONE JSON line from rank 0,
whole-job cell counts, the scaling mode."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def run_ranks(n, port, *extra):
    port = free_port()                                  # (the argument is kept for readability of the call sites)
    env = dict(os.environ, MOSAIC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(n), "--steps", "2", "--warmup", "1", *extra]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]            # one JSON line, nothing else on stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_weak_bands_of_config_b():
    single = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "B", "--steps", "1",
                                        "--warmup", "0", "--no-cpu"], cwd=ROOT, capture_output=True, text=True,
                                       timeout=300, check=True).stdout.strip().splitlines()[-1])
    out = run_ranks(2, 29541, "--config", "B")       # weak bands need square cells: B (config D is the same path, 25 s)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 2
    assert out["config"]["cells"] == 2 * single["config"]["cells"]          # one band per rank
    assert "REHEARSAL" in out["data"] and out["value"] > 0
    assert out["last_step_device_ms"]["band_alarms"] == 0


@pytest.mark.gpu
def test_three_ranks_strong_shards_of_config_c():
    out = run_ranks(3, 29542, "--config", "C", "--scaling", "strong")
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert "x3 (strong)" in out["config"]["parallelism"]
    assert out["value"] > 0
