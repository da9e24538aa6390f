"""bench.py's host-side helpers (no GPU): the parity sample, the result digest, the rehearsal configurations."""
import numpy as np

import bench
import cases
from mosaicmagnifique_amd import GridUtility as G


def cells_of(shape, rows, cols):
    group = cases.group_of(shape)
    state = G.getGridState(group, rows, cols)[0]
    return G.cell_records(group, state, rows, cols)


def test_parity_sample_with_a_repeat_penalty_is_the_first_two_grid_rows():
    shape = cases.square_shape(64)
    cells = cells_of(shape, 1024, 1024)
    sample, text = bench.parity_sample(dict(rep=(4, 500), size=64, shape=shape), cells)
    assert len(sample) == 32 and np.array_equal(sample, cells[:32])          # 16 cells per row
    assert "2 full grid rows" in text


def test_parity_sample_without_a_penalty_is_random_and_covers_clipped_cells():
    shape = cases.hexagon_shape()
    cells = cells_of(shape, 1024, 1024)
    w = dict(rep=(0, 0), size=64, shape=shape)
    sample, text = bench.parity_sample(w, cells)
    again, _ = bench.parity_sample(w, cells)
    assert np.array_equal(sample, again)                                      # seeded
    whole = (sample["row_min"] == 0) & (sample["row_max"] == 64) & (sample["col_min"] == 0) & (sample["col_max"] == 64)
    assert whole.sum() == 64 and (~whole).sum() == 16 and len(np.unique(sample[["grid_x", "grid_y"]])) == 80
    masks = shape.masks4() != 0
    for c in sample[~whole]:                                                  # clipped ones hold active pixels
        assert masks[c["mask_id"]][c["row_min"]:c["row_max"], c["col_min"]:c["col_max"]].any()
    assert (text % len(sample)).startswith("80 cells drawn at random")
    # raster order is kept (the oracle walks them with the engine's earlier picks in place)
    order = sample["grid_y"].astype(np.int64) * 10000 + sample["grid_x"]
    assert (np.diff(order) > 0).all()


def test_result_digest_sees_every_bit():
    best = np.arange(12, dtype=np.int64).reshape(3, 4)
    score = np.linspace(0.0, 1.0, 12).reshape(3, 4)
    score[0, 0] = np.nan
    d = bench.result_digest(best, score)
    assert d == bench.result_digest(best.copy(), score.copy()) and len(d) == 64
    b2 = best.copy(); b2[2, 3] += 1
    s2 = score.copy(); s2[1, 1] = np.nextafter(s2[1, 1], 2.0)
    assert bench.result_digest(b2, score) != d and bench.result_digest(best, s2) != d


def test_rehearsal_configurations_keep_the_shape_of_d_and_e():
    d, ds = bench.CONFIGS["D"], bench.CONFIGS["Ds"]
    assert ds[2:4] == d[2:4] and ds[5:8] == d[5:8]                           # cell, shape, type, repeat range and addition
    e, es = bench.CONFIGS["E"], bench.CONFIGS["Es"]
    assert es[2:4] == e[2:4] and es[5:8] == e[5:8]
    cells = cells_of(cases.square_shape(48), es[0], es[1])
    assert len(cells) == 441 and (cells["row_max"] - cells["row_min"]).min() == 40      # clipped last row / column


def test_weak_bands_stack_the_single_gpu_grid():
    w1 = bench.build_workload("Ds", 1)
    w3 = bench.build_workload("Ds", 1, "weak", bands=3)
    assert len(w3["cells"]) == 3 * len(w1["cells"]) and w3["grid_w"] == w1["grid_w"]
    assert w3["grid_h"] == 3 * (w1["grid_h"] - G.PAD_GRID) + G.PAD_GRID
    assert np.array_equal(w3["cells"]["y"][:len(w1["cells"])], w1["cells"]["y"])


def test_gpus_n_without_a_launcher_starts_the_ranks_itself(monkeypatch):
    """`python3 bench.py --gpus N` as the driver runs it (no WORLD_SIZE around it): the ranks are started as a CHILD
    `python -m torch.distributed.run` (never an exec, nothing of this process has touched the GPU), rendezvous on
    127.0.0.1, and rank 0's one JSON line is passed on; the launcher's failure is this process's failure."""
    import subprocess
    import sys
    import types

    import pytest

    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return types.SimpleNamespace(returncode=seen.get("rc", 0), stdout="NCCL banner\n{\"metric\": \"m\", \"n_gpus\": 2}\n")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("MOSAIC_BENCH_REHEARSAL", "1")          # (no device count needed on this box)
    out = []
    bench.launch_ranks(2, out.append, argv=["--gpus", "2", "--steps", "3", "--warmup", "1"])
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert out == ['{"metric": "m", "n_gpus": 2}']
    seen["rc"] = 3
    with pytest.raises(SystemExit) as err:
        bench.launch_ranks(2, out.append, argv=["--gpus", "2"])
    assert err.value.code == 3


def test_parity_sample_moves_to_interior_rows_once_the_golden_vouches_for_the_grid():
    """With the whole grid equal to the committed CPU golden the induction over the raster order has its base anywhere: the
    sample is two full grid rows at a seeded interior offset, with the whole rows-above window of calculateRepeats in play."""
    shape = cases.square_shape(64)
    cells = cells_of(shape, 1024, 1024)                                          # 16 x 16 cells
    w = dict(rep=(4, 500), size=64, shape=shape, L=5000, golden_ok=True)
    sample, text = bench.parity_sample(w, cells)
    rows = np.unique(sample["grid_y"])
    assert len(sample) == 32 and len(rows) == 2 and rows[1] == rows[0] + 1
    assert rows[0] - cells["grid_y"].min() >= 5 and "interior" in text          # at least range + 1 rows above
    again, _ = bench.parity_sample(w, cells)
    assert np.array_equal(sample, again)
