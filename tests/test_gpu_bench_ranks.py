"""bench.py's N > 1 code path on a one-GPU box: the ranks share GPU 0 and the collective runs over gloo
(MOSAIC_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device).  The timings of such a run are void; what is
checked is (a) the launch contract -- torch.distributed.run with N ranks, ONE JSON line from rank 0, whole-job cell
counts, the scaling mode -- and (b) the RESULT: the sha256 of the best-fit grid and the scores an N-rank run of the
real engine through mosaicmagnifique_amd/dist.py ends with (`result_digest`) equals the one-GPU run's, for weak
bands and strong shards, with the deepening rounds forced (spec_k = 2).  A GPU box admits six processes on its
card (this process holds it too), so the rank counts here are 2, 3 and 4; eight shards run in ONE process through the C ABI's group
(tests/test_gpu_group.py)."""
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


def last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout[-2000:]              # one JSON line, nothing else on stdout
    return json.loads(lines[0])


def run_single(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu", *extra],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return last_json(p.stdout)


def run_ranks(n, *extra):
    env = dict(os.environ, MOSAIC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(n), "--steps", "2", "--warmup", "1", *extra]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return last_json(p.stdout)


@pytest.mark.gpu
def test_two_ranks_weak_bands_of_config_b():
    single = run_single("--config", "B", "--bands", "2")          # the same 2-band mosaic on one GPU
    out = run_ranks(2, "--config", "B")       # weak bands need square cells: B (config D is the same path, 25 s)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 2
    gold = out["parity_on_golden"]            # no repeat penalty: BOTH bands are the golden's job
    assert gold["indices_equal"] is True and gold["cells"] == 2 * 1024 and "all 2" in gold["bands_compared"]
    assert out["config"]["cells"] == single["config"]["cells"] == 2 * 1024          # one band per rank
    assert "REHEARSAL" in out["data"] and out["value"] > 0
    assert out["last_step_device_ms"]["band_alarms"] == 0
    assert out["result_digest"] == single["result_digest"]


@pytest.mark.gpu
def test_three_ranks_strong_shards_of_config_c():
    single = run_single("--config", "C")
    out = run_ranks(3, "--config", "C", "--scaling", "strong")
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert "x3 (strong)" in out["config"]["parallelism"]
    assert out["value"] > 0
    assert out["result_digest"] == single["result_digest"]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 4])
def test_weak_bands_with_the_repeat_penalty_and_forced_deepening(n):
    """The reduced shape of config D (64 x 64 cells, CIEDE2000, repeat range 4 + 500): the repeat window reaches across
    the band seams, the short lists of 2 leave picks unverified, their owners deepen them and the lists are gathered
    again -- and the N-band grid equals the one a single GPU gives for the same mosaic."""
    single = run_single("--config", "Ds", "--bands", str(n), "--opt", "spec_k=2")
    out = run_ranks(n, "--config", "Ds", "--opt", "spec_k=2")
    assert out["n_gpus"] == n and out["scaling"] == "weak"
    assert out["config"]["cells"] == single["config"]["cells"] == n * 256
    assert out["result_digest"] == single["result_digest"]
    # the N-rank line checks ITSELF (round 6): band 0 against the committed CPU golden of the reduced config D, and two full grid rows
    # inside the LAST band evaluated by the CPU oracle as the raster loop reaches them (the window of calculateRepeats in play)
    gold = out["parity_on_golden"]
    assert gold["indices_equal"] is True and gold["cells"] == 256 and gold["max_rel_score_err"] < 1e-9 and "band 0" in gold["bands_compared"]
    par = out["parity_on_cpu_sample"]
    assert par["indices_equal"] is True and par["cells"] == 32 and par["max_rel_score_err"] < 1e-9
    assert out["rccl_ranks_seen"] == n and len(out["dist_phase_ms"]["score_per_rank"]) == n


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4])
def test_strong_shards_of_the_reduced_config_e(n):
    """The reduced shape of config E (48 x 48 cells, last row / column clipped, no repeat penalty): ceil(441 / N)
    contiguous shards, the last one short."""
    single = run_single("--config", "Es")
    out = run_ranks(n, "--config", "Es")           # E-shaped configs default to strong scaling
    assert out["config"]["cells"] == single["config"]["cells"] == 441
    assert out["result_digest"] == single["result_digest"]
    # ... and against the committed CPU golden of the whole grid, inside the N-rank job itself (tests/golden/grid_Es.npz)
    gold = out["parity_on_golden"]
    assert gold["indices_equal"] is True and gold["cells"] == 441 and gold["cells_differing"] == 0 and gold["max_rel_score_err"] < 1e-9
    assert out["rccl_ranks_seen"] == n and "gloo" in out["collective_backend"]
    assert len(out["dist_phase_ms"]["score_per_rank"]) == n and all(v > 0 for v in out["dist_phase_ms"]["score_per_rank"])


@pytest.mark.gpu
def test_plain_gpus_2_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher around it -- the form the driver uses for N = 1 -- starts the two
    ranks itself as a child torch.distributed.run, exits 0 with rank 0's single JSON line, and (rehearsal: both ranks on
    GPU 0 over gloo) reproduces the digest of the one-GPU run of the same two-band mosaic."""
    single = run_single("--config", "B", "--bands", "2")
    env = dict(os.environ, MOSAIC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "B"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["cells"] == 2 * 1024
    assert out["result_digest"] == single["result_digest"]
    # the phases of the sharded step, per-rank maxima: what makes the first hardware curve diagnosable
    ph = out["dist_phase_ms"]
    assert set(ph) >= {"score", "allgather", "resolve", "deepen"} and ph["score"] > 0 and ph["resolve"] > 0
