"""Offline simulation (numpy + the oracle): what a FINER block of the early-exit bound would buy.  The kernel's bound sums
|sum over a 64-position block of (cell - candidate)| per channel; the zero-mean part of the differences (noise) cancels
inside a block in proportion to 1 / sqrt(block size), so smaller blocks capture more of it.  For a few cells of a bench
configuration and blocks of 64 / 32 / 16 / 8 positions (contiguous runs of the plane order): the share of candidates the
bound alone drops, and the groups the survivors walk (true fp64 sums of the visited groups + bound of the others against
the limit), in the kernel's visiting order (largest group bounds first).
Not a test: a design tool (python tools/block_size_sim.py [config] [n_cells] [K])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_ffi  # noqa: E402
from test_filter_bound import rotation_bound, sqrt_ratio7  # noqa: E402
from bound_sim import walk_order  # noqa: E402

import bench  # noqa: E402


def block_bound(cell, lib, bs):
    """cell [P][3], lib [L][P][3] in plane order -> bound per block [L][P / bs]."""
    L, P, _ = lib.shape
    nb = P // bs
    cb, lb = cell.reshape(nb, bs, 3), lib.reshape(L, nb, bs, 3)
    dev = 0.5 * (np.abs(cb[..., 0] - 50).max(-1)[None] + np.abs(lb[..., 0] - 50).max(-1))
    s_l = 1.0 + 0.015 * dev ** 2 / np.sqrt(20.0 + dev ** 2)
    cbar = 0.5 * (np.hypot(cb[..., 1], cb[..., 2]).max(-1)[None] + np.hypot(lb[..., 1], lb[..., 2]).max(-1))
    gc = (1.5 - 0.5 * sqrt_ratio7(cbar)) * cbar
    s_c = 1.0 + 0.045 * gc
    e_max = rotation_bound(np.broadcast_to(cb[None], lb.shape), lb)
    half_rt = sqrt_ratio7(gc) * np.sin(np.radians(60.0) * e_max) * 1.0001
    s_h = 1.0 + 0.015 * 1.5726 * gc
    a, c, b = 1.0 / s_c ** 2, 1.0 / s_h ** 2, half_rt / (s_c * s_h)
    lam = np.maximum(0.5 * (a + c) - np.sqrt(0.25 * (a - c) ** 2 + b ** 2), 0.0)
    gmin = 1.5 - 0.5 * sqrt_ratio7(cbar)
    d = np.abs(cb.sum(1)[None] - lb.sum(2))
    return np.sqrt((d[..., 0] / s_l) ** 2 + lam * ((gmin * d[..., 1]) ** 2 + d[..., 2] ** 2))


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "D"
    n_cells = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    w = bench.build_workload(config, 1)
    o = oracle_ffi.load()
    size, L = w["size"], w["L"]
    P = size * size
    ng = P // 256
    order = walk_order(size)
    lib = w["lib"].reshape(L, P, 3)[:, order].astype(np.float64)
    rng = np.random.default_rng(1)
    cells = w["cells_single"]
    pick = rng.choice(len(cells), n_cells, replace=False)
    sizes = (64, 32, 16, 8)
    dropped = {bs: 0 for bs in sizes}
    walked = {bs: 0.0 for bs in sizes}
    tight = {bs: [] for bs in sizes}
    for ci in pick:
        r = cells[ci]
        cell = w["img"][r["y"]:r["y"] + size, r["x"]:r["x"] + size].reshape(-1, 3)[order].astype(np.float64)
        grp = {bs: block_bound(cell, lib, bs).reshape(L, ng, 256 // bs).sum(2) for bs in sizes}      # [L][ng]
        # the limit: K-th lowest true score among the most promising candidates (by the finest bound)
        cand = np.argsort(grp[8].sum(1))[:200]
        cellf = np.broadcast_to(cell.astype(np.float32)[None], (len(cand), P, 3))
        tv = o.colour_diff_batch(2, cellf.reshape(-1, 3), lib[cand].astype(np.float32).reshape(-1, 3)).reshape(len(cand), -1).sum(1)
        limit = np.sort(tv)[K - 1] * 1.0003 + 8.0
        # true group sums of a sample of the candidates (all of those any bound keeps would be too many for the oracle)
        samp = rng.choice(L, 600, replace=False)
        cellf = np.broadcast_to(cell.astype(np.float32)[None], (len(samp), P, 3))
        ta = o.colour_diff_batch(2, cellf.reshape(-1, 3), lib[samp].astype(np.float32).reshape(-1, 3)).reshape(len(samp), ng, 256).sum(2)
        for bs in sizes:
            g = grp[bs][samp]
            assert (g <= ta * (1 + 1e-9) + 1e-9).all()
            alive = g.sum(1) <= limit
            dropped[bs] += int((~alive).sum())
            tight[bs].append(np.median(g.sum(1) / ta.sum(1)))
            # visiting order: descending group bound
            idx = np.argsort(-g, axis=1, kind="stable")
            gs, ts = np.take_along_axis(g, idx, 1), np.take_along_axis(ta, idx, 1)
            part = np.concatenate([np.zeros((len(samp), 1)), np.cumsum(ts, 1)], 1)           # true sums of the first j groups
            rest = np.concatenate([np.cumsum(gs[:, ::-1], 1)[:, ::-1], np.zeros((len(samp), 1))], 1)   # bound of groups >= j
            stop = (part + rest) > limit
            first = np.where(stop.any(1), stop.argmax(1), ng)
            walked[bs] += float(first[alive].sum())
        print("cell %d: limit %.0f" % (ci, limit), " ".join("bs%d: bound/true %.3f" % (bs, tight[bs][-1]) for bs in sizes), flush=True)
    n = n_cells * 600
    for bs in sizes:
        print("block %2d: dropped by the bound alone %.3f of candidates; groups walked %.4f of all; bound / true (median) %.3f" % (
            bs, dropped[bs] / n, walked[bs] / (n * ng), float(np.mean(tight[bs]))))


if __name__ == "__main__":
    main()
