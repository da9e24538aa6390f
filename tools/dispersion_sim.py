"""Offline simulation (numpy + the oracle): a DISPERSION term in the early-exit bound.

The block bound is  sum_p dE_p >= || sum_p v_p ||  with v_p = (dL / S_L, sqrt(lambda) g_min da, sqrt(lambda) db) at the block's worst
constants (filter.hip stats_bound).  Taking absolute values componentwise first keeps every norm:  sum_p ||v_p|| >= || sum_p |v_p| ||,
and per component k two lower bounds of  sum_p |c_p - l_p|  hold:
    | sum_p c_p - sum_p l_p |                              (what the kernel uses: the means differ)
    | D_l - D_c |,  D_x = min_m sum_p |x_p - m|            (the dispersions about the medians differ: for any m,
                                                            sum |c - l| >= sum |l - m| - sum |c - m|, take m = med_c or med_l)
so the component may take the LARGER of the two.  Noise on one side that the other side lacks (library patches of +-32 against a
smooth target, a textured tile against a flat cell) cancels in the sums and is invisible to the first bound; the second sees it.
Three more statistics per image and block.  This tool: config D's content (or any bench configuration), a few cells x 600
candidates, true fp64 group sums from the oracle, the kernel's visiting order -- candidates dropped by the bound alone, groups
walked, bound / true -- for the current bound and the one with the dispersion term.
Not a test: a design tool (python tools/dispersion_sim.py [config] [n_cells] [K])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_ffi  # noqa: E402
from test_filter_bound import rotation_bound, sqrt_ratio7  # noqa: E402
from bound_sim import walk_order  # noqa: E402

import bench  # noqa: E402


def block_bounds(cell, lib, bs=64):
    """cell [P][3], lib [L][P][3] in plane order -> (current bound, bound with the dispersion term) per block [L][P / bs]."""
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
    d = np.abs(cb.sum(1)[None] - lb.sum(2))                                     # [L][nb][3]
    disp_c = np.abs(cb - np.median(cb, axis=1, keepdims=True)).sum(1)           # [nb][3]
    disp_l = np.abs(lb - np.median(lb, axis=2, keepdims=True)).sum(2)           # [L][nb][3]
    dd = np.maximum(d, np.abs(disp_l - disp_c[None]))
    cur = np.sqrt((d[..., 0] / s_l) ** 2 + lam * ((gmin * d[..., 1]) ** 2 + d[..., 2] ** 2))
    new = np.sqrt((dd[..., 0] / s_l) ** 2 + lam * ((gmin * dd[..., 1]) ** 2 + dd[..., 2] ** 2))
    return cur, new


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
    whole = cells[(cells["row_max"] - cells["row_min"] == size) & (cells["col_max"] - cells["col_min"] == size)]
    pick = rng.choice(len(whole), n_cells, replace=False)
    names = ("current", "with dispersion")
    dropped = {k: 0 for k in names}
    walked = {k: 0.0 for k in names}
    tight = {k: [] for k in names}
    n_samp = min(600, L)
    for ci in pick:
        r = whole[ci]
        cell = w["img"][r["y"]:r["y"] + size, r["x"]:r["x"] + size].reshape(-1, 3)[order].astype(np.float64)
        cur, new = block_bounds(cell, lib)
        grp = {"current": cur.reshape(L, ng, 4).sum(2), "with dispersion": new.reshape(L, ng, 4).sum(2)}
        cand = np.argsort(grp["with dispersion"].sum(1))[:200]
        cellf = np.broadcast_to(cell.astype(np.float32)[None], (len(cand), P, 3))
        tv = o.colour_diff_batch(2, cellf.reshape(-1, 3), lib[cand].astype(np.float32).reshape(-1, 3)).reshape(len(cand), -1).sum(1)
        limit = np.sort(tv)[K - 1] * 1.0003 + 8.0
        samp = rng.choice(L, n_samp, replace=False)
        cellf = np.broadcast_to(cell.astype(np.float32)[None], (len(samp), P, 3))
        ta = o.colour_diff_batch(2, cellf.reshape(-1, 3), lib[samp].astype(np.float32).reshape(-1, 3)).reshape(len(samp), ng, 256).sum(2)
        for k in names:
            g = grp[k][samp]
            assert (g <= ta * (1 + 1e-9) + 1e-9).all(), (k, float((g / np.maximum(ta, 1e-9)).max()))
            alive = g.sum(1) <= limit
            dropped[k] += int((~alive).sum())
            tight[k].append(np.median(g.sum(1) / ta.sum(1)))
            idx = np.argsort(-g, axis=1, kind="stable")
            gs, ts = np.take_along_axis(g, idx, 1), np.take_along_axis(ta, idx, 1)
            part = np.concatenate([np.zeros((len(samp), 1)), np.cumsum(ts, 1)], 1)
            rest = np.concatenate([np.cumsum(gs[:, ::-1], 1)[:, ::-1], np.zeros((len(samp), 1))], 1)
            stop = (part + rest) > limit
            first = np.where(stop.any(1), stop.argmax(1), ng)
            walked[k] += float(first[alive].sum())
        print("cell %d: limit %.0f" % (ci, limit), " ".join("%s: bound/true %.3f" % (k, tight[k][-1]) for k in names), flush=True)
    n = n_cells * n_samp
    for k in names:
        print("%-16s dropped by the bound alone %.3f of candidates; groups walked %.4f of all; bound / true (median) %.3f" % (
            k, dropped[k] / n, walked[k] / (n * ng), float(np.mean(tight[k]))))


if __name__ == "__main__":
    main()
