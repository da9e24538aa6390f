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


@pytest.mark.parametrize("n,k", [(1024, 2), (4097, 42), (5, 3)])
def test_allgather_of_candidate_lists_world2_gloo(tmp_path, n, k):
    world = 2
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
