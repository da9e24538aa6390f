"""The N > 1 path on CPU: world_size-2 gloo processes exercise the shard split, the
padding and the all-gather of the candidate lists (mosaicmagnifique_amd.dist) --
everything of the multi-GPU step except the device kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from mosaicmagnifique_amd.dist import (allgather_candidates, merge_on_host, padded_len,  # noqa: E402
                                       shard_range)


def test_shard_ranges_tile_the_cells():
    for n in (0, 1, 7, 8, 9, 4096, 29241):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            per = padded_len(n, world)
            assert all(hi - lo <= per for lo, hi in spans) and per * world >= n


def _fake_lists(n, k):
    rng = np.random.default_rng(99)
    idx = rng.integers(0, 5000, (n, k)).astype(np.int32)
    score = np.sort(rng.uniform(1e4, 1e5, (n, k)), axis=1)
    return idx, score


def _worker(rank, world, port, n, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx, score = _fake_lists(n, k)                      # the "true" global lists
    lo, hi = shard_range(n, world, rank)
    per = padded_len(n, world)
    li = torch.full((per, k), -1, dtype=torch.int32)
    ls = torch.full((per, k), float("inf"), dtype=torch.float64)
    li[:hi - lo] = torch.from_numpy(idx[lo:hi])         # this rank "scored" its own block
    ls[:hi - lo] = torch.from_numpy(score[lo:hi])
    gi, gs = allgather_candidates(li, ls, n, world)
    np.save(os.path.join(out_dir, "idx%d.npy" % rank), gi.numpy())
    np.save(os.path.join(out_dir, "score%d.npy" % rank), gs.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n,k", [(2, 1024, 2), (2, 4097, 42), (2, 5, 3),
                                       (8, 4097, 42), (8, 5, 3), (3, 29, 5)])    # (eight ranks: the node the scaling run uses; fewer cells than ranks: empty shards)
def test_allgather_of_candidate_lists_world2_gloo(tmp_path, world, n, k):
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path)), nprocs=world, join=True)
    idx, score = _fake_lists(n, k)
    for rank in range(world):
        gi = np.load(tmp_path / ("idx%d.npy" % rank))
        gs = np.load(tmp_path / ("score%d.npy" % rank))
        assert gi.shape == (n, k) and np.array_equal(gi, idx)      # every rank holds the full, ordered lists
        assert np.array_equal(gs, score)


def test_merge_on_host_matches_allgather_layout():
    n, k, world = 11, 4, 3
    idx, score = _fake_lists(n, k)
    per = padded_len(n, world)
    shards = []
    for r in range(world):
        lo, hi = shard_range(n, world, r)
        si = np.full((per, k), -1, np.int32); ss = np.full((per, k), np.inf)
        si[:hi - lo], ss[:hi - lo] = idx[lo:hi], score[lo:hi]
        shards.append((si, ss))
    # padding rows of a short last shard sit at the very end, so trimming is enough
    mi, ms = merge_on_host(shards, n)
    assert np.array_equal(mi, idx) and np.array_equal(ms, score)


# ---------------------------------------------------------------------------------------------
# The whole sharded step (ShardedBestFit.generate: speculative short lists, all-gather, verifying
# resolve, deepening of unverified cells by their owner, second gather) with world size 2 on CPU.
# The device engine is replaced by a stand-in with the same four methods, computing exact lists
# with numpy and resolving with the oracle's repeat window; the result must equal the oracle's
# generateBestFits on the same inputs.
# ---------------------------------------------------------------------------------------------
import ctypes  # noqa: E402


class StandInEngine:
    """Engine look-alike on CPU tensors: base scores by brute force (RGB-Euclid, fp64)."""
    torch_device = torch.device("cpu")
    device = 0

    def __init__(self, oracle, shape, img, lib, cells):
        from oracle_ffi import shape_of
        self.o, self.n_library = oracle, len(lib)
        masks = shape.masks4()
        size = shape.getSize()
        base = np.empty((len(cells), len(lib)))
        for i, r in enumerate(cells):
            cell, _ = oracle.cell_at(shape_of(shape), int(r["grid_x"]) - 2, int(r["grid_y"]) - 2, img)
            active = masks[r["mask_id"]] != 0
            box = np.zeros((size, size), bool)
            box[r["row_min"]:r["row_max"], r["col_min"]:r["col_max"]] = True
            active &= box
            d = np.sqrt(((lib.astype(np.float64) - cell.astype(np.float64)[None]) ** 2).sum(-1))
            base[i] = (d * active[None]).reshape(len(lib), -1).sum(1)
        self.base, self.cells = base, cells
        self.calls = []

    def candidates_per_cell(self, rng, add):
        return min(self.n_library, rng * (2 * rng + 1) + rng + 2 if rng > 0 and add > 0 else 2)

    def speculative_candidates(self, rng, add, spec_k):
        k = self.candidates_per_cell(rng, add)
        return spec_k if 0 < spec_k < k - 2 else k

    @staticmethod
    def _view(ptr, shape, dtype):
        n = int(np.prod(shape))
        buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def score_cells(self, cells, diff_type, k, out_index=None, out_score=None, on_device=False, k_fill=None):
        k_fill = k if k_fill is None else k_fill
        rows = [int(np.nonzero((self.cells["grid_x"] == c["grid_x"]) & (self.cells["grid_y"] == c["grid_y"]))[0][0])
                for c in cells]
        idx = self._view(out_index, (len(cells), k), np.int32)
        sc = self._view(out_score, (len(cells), k), np.float64)
        idx[:], sc[:] = -1, np.inf
        for j, r in enumerate(rows):
            order = np.lexsort((np.arange(self.n_library), self.base[r]))[:k_fill]
            idx[j, :len(order)], sc[j, :len(order)] = order, self.base[r][order]
        self.calls.append(("score", len(cells), k_fill))

    def resolve(self, cells, grid_w, grid_h, diff_type, rng, add, k, cand_index, cand_score, on_device=False,
                deep=None, k_fill=None):
        n = len(cells)
        idx = self._view(cand_index, (n, k), np.int32)
        sc = self._view(cand_score, (n, k), np.float64)
        grid = np.full((grid_h, grid_w), -1, np.int64)
        best = np.full((grid_h, grid_w), self.n_library, np.int64)
        score = np.full((grid_h, grid_w), np.nan)
        bad = np.zeros(n, np.uint8)
        for i, c in enumerate(cells):
            kk = k if deep is None or deep[i] else k_fill
            rep = self.o.repeats(grid, int(c["grid_x"]), int(c["grid_y"]), rng, add) if rng > 0 else {}
            valid = [(sc[i, j] + rep.get(int(idx[i, j]), 0), int(idx[i, j])) for j in range(kk) if idx[i, j] >= 0]
            total, pick = min(valid)
            grid[c["grid_y"], c["grid_x"]] = best[c["grid_y"], c["grid_x"]] = pick
            score[c["grid_y"], c["grid_x"]] = total
            if deep is not None and not deep[i]:
                bad[i] = not (total < sc[i, kk - 1])
        self.calls.append(("resolve", int(bad.sum())))
        return (best, score) if deep is None else (best, score, bad)


def _sharded_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import oracle_ffi
    from mosaicmagnifique_amd import GridUtility as G
    from mosaicmagnifique_amd.dist import ShardedBestFit
    oracle = oracle_ffi.load()
    shape = cases.square_shape(8)
    group = cases.group_of(shape)
    img, lib = cases.inputs(52, 67, 8, 60, 0, lattice=40)              # smooth target: neighbours collide
    state = G.getGridState(group, 52, 67)[0]
    cells = G.cell_records(group, state, 52, 67)
    engine = StandInEngine(oracle, shape, img, lib, cells)
    best, score = ShardedBestFit(engine, rank, world).generate(cells, state.shape[1], state.shape[0], 0, 2, 2000, spec_k=3)
    np.save(os.path.join(out_dir, "best%d.npy" % rank), best)
    np.save(os.path.join(out_dir, "calls%d.npy" % rank), np.array([c[1] for c in engine.calls if c[0] == "resolve"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])       # (63 cells: shards of 32 / 31, 21 / 21 / 21, and seven of 8 with a last one of 7)
def test_sharded_speculate_verify_deepen_world2_gloo(tmp_path, world):
    mp.spawn(_sharded_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import oracle_ffi
    oracle = oracle_ffi.load()
    shape = cases.square_shape(8)
    img, lib = cases.inputs(52, 67, 8, 60, 0, lattice=40)
    want, _, _ = oracle.generate_best_fits(oracle_ffi.shape_of(shape), img, lib, shape.masks4(), 0, 2, 2000)
    for rank in range(world):
        best = np.load(tmp_path / ("best%d.npy" % rank))
        assert np.array_equal(np.where(best == len(lib), -1, best), want)
        unverified = np.load(tmp_path / ("calls%d.npy" % rank))
        assert len(unverified) >= 2 and unverified[0] > 0 and unverified[-1] == 0     # deepening happened, then settled


def _diverging_worker(rank, world, port, out_dir):
    """Rank 1's stand-in engine vouches for every pick (reports no unverified cell) while rank 0's does
    not: without the cross-check the ranks would enter different rounds and the next all-gather would
    hang; with it BOTH ranks raise."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import oracle_ffi
    from mosaicmagnifique_amd import GridUtility as G
    from mosaicmagnifique_amd.dist import ShardedBestFit
    oracle = oracle_ffi.load()
    shape = cases.square_shape(8)
    group = cases.group_of(shape)
    img, lib = cases.inputs(52, 67, 8, 60, 0, lattice=40)
    state = G.getGridState(group, 52, 67)[0]
    cells = G.cell_records(group, state, 52, 67)
    engine = StandInEngine(oracle, shape, img, lib, cells)
    if rank == 1:
        honest = engine.resolve

        def lenient(*a, **kw):
            out = honest(*a, **kw)
            return out if len(out) == 2 else (out[0], out[1], np.zeros_like(out[2]))
        engine.resolve = lenient
    verdict = "no error"
    try:
        ShardedBestFit(engine, rank, world).generate(cells, state.shape[1], state.shape[0], 0, 2, 2000, spec_k=3)
    except RuntimeError as err:
        verdict = str(err)
    open(os.path.join(out_dir, "verdict%d.txt" % rank), "w").write(verdict)
    dist.destroy_process_group()


def test_ranks_that_disagree_raise_instead_of_hanging_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_diverging_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert "ranks disagree" in open(tmp_path / ("verdict%d.txt" % rank)).read()
