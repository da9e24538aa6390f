"""What an anisotropic block bound (hue differences credited with 1/S_H^2 - 1/S_C^2 more) would buy on config D: visits of the
filter cascade with (cur) today's block bound, (v1) + a rigorous hue term from the arcs' gap and the blocks' minimum chroma,
(v2) + the unattainable ideal: the true sum of dH' of the block pair."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
sys.argv = ['x']
import oracle_ffi
from test_filter_bound import rotation_bound, sqrt_ratio7, hue_arcs, wrap180
import bench
from tools_bound_sim import walk_order

def main(n_cells=2, K=4):
    w = bench.build_workload('D', 1)
    o = oracle_ffi.load()
    size, L = w["size"], w["L"]
    Lsub = 1500
    order = walk_order(size)
    rng = np.random.default_rng(1)
    sub = np.sort(rng.choice(L, Lsub, replace=False))
    lib = w["lib"].reshape(L, size * size, 3)[sub][:, order].astype(np.float64)
    nb = size * size // 64
    lb = lib.reshape(Lsub, nb, 64, 3)
    cells = w["cells_single"]
    pick = rng.choice(len(cells), n_cells, replace=False)
    res = {}
    for ci in pick:
        r = cells[ci]
        cell = w["img"][r["y"]:r["y"] + size, r["x"]:r["x"] + size].reshape(-1, 3)[order].astype(np.float64)
        cb = cell.reshape(nb, 64, 3)
        # constants
        dev = 0.5 * (np.abs(cb[..., 0] - 50).max(-1)[None] + np.abs(lb[..., 0] - 50).max(-1))
        s_l = 1.0 + 0.015 * dev ** 2 / np.sqrt(20.0 + dev ** 2)
        c_cell, c_lib = np.hypot(cb[..., 1], cb[..., 2]), np.hypot(lb[..., 1], lb[..., 2])
        cbar = 0.5 * (c_cell.max(-1)[None] + c_lib.max(-1))
        gmin = 1.5 - 0.5 * sqrt_ratio7(cbar)
        gc = gmin * cbar
        s_c = 1.0 + 0.045 * gc
        e_max = rotation_bound(np.broadcast_to(cb[None], lb.shape), lb)
        half_rt = sqrt_ratio7(gc) * np.sin(np.radians(60.0) * e_max) * 1.0001
        s_h = 1.0 + 0.015 * 1.5726 * gc
        a, c, b = 1.0 / s_c ** 2, 1.0 / s_h ** 2, half_rt / (s_c * s_h)
        lam = np.maximum(0.5 * (a + c) - np.sqrt(0.25 * (a - c) ** 2 + b ** 2), 0.0)
        d = np.abs(cb.sum(1)[None] - lb.sum(2))
        iso2 = (d[..., 0] / s_l) ** 2 + lam * ((gmin * d[..., 1]) ** 2 + d[..., 2] ** 2)
        # anisotropic extra: q >= (a - |b|) dC'^2 + (c - |b|) dH'^2 ; extra weight on dH'^2 beyond lam
        w_c, w_h = np.maximum(a - np.abs(b), 0), np.maximum(c - np.abs(b), 0)
        extra_w = np.maximum(w_h - np.maximum(lam, w_c), 0.0)       # conservative: iso part keeps lam, the hue part adds (w_h - max(lam, w_c))
        c1, w1 = hue_arcs(np.broadcast_to(cb[None], lb.shape)); c2, w2 = hue_arcs(lb)
        D = np.abs(wrap180(c2 - c1))
        gap = np.maximum(0.0, D - w1 - w2)
        gap = np.where(D + w1 + w2 >= 180.0, np.minimum(gap, np.maximum(0.0, 360.0 - D - w1 - w2)), gap)
        gap = np.minimum(gap, 180.0)
        sin_half = np.sin(np.radians(gap) / 2)
        h_v1 = 64 * 2 * np.sqrt(c_cell.min(-1)[None] * c_lib.min(-1)) * sin_half
        # ideal: true sum of dH' per block pair (G = gmin scaling ignored: a' = a here -> slight underestimate is fine for a what-if)
        h1 = np.arctan2(cb[..., 2], cb[..., 1])[None]; h2 = np.arctan2(lb[..., 2], lb[..., 1])
        dh = np.abs(np.angle(np.exp(1j * (h2 - h1))))
        h_v2 = (2 * np.sqrt(c_cell[None] * c_lib) * np.sin(dh / 2)).sum(-1)
        blk = {"cur": np.sqrt(iso2), "v1": np.sqrt(iso2 + extra_w * h_v1 ** 2), "v2": np.sqrt(iso2 + extra_w * h_v2 ** 2)}
        # true group sums
        cellf = np.broadcast_to(cell.astype(np.float32)[None], (Lsub, size * size, 3))
        tv = o.colour_diff_batch(2, np.ascontiguousarray(cellf.reshape(-1, 3)), np.ascontiguousarray(lib.astype(np.float32).reshape(-1, 3))).reshape(Lsub, nb // 4, 256).sum(2)
        total = tv.sum(1)
        limit = np.sort(total)[K - 1] * 1.0003 + 8.0
        ng = nb // 4
        for k, bl in blk.items():
            grp = bl.reshape(Lsub, ng, 4).sum(2)
            assert (grp <= tv * (1 + 1e-9) + 1e-6).all() or k == "v2", k
            # descending-bound order
            orderg = np.argsort(-grp, axis=1, kind="stable")
            g_s = np.take_along_axis(grp, orderg, 1); t_s = np.take_along_axis(tv, orderg, 1)
            rest = np.concatenate([np.cumsum(g_s[:, ::-1], 1)[:, ::-1], np.zeros((Lsub, 1))], 1)
            part = np.concatenate([np.zeros((Lsub, 1)), np.cumsum(t_s, 1)], 1)
            stop = (part + rest) > limit
            first = np.where(stop.any(1), stop.argmax(1), ng)
            dropped = (rest[:, 0] > limit).mean()
            rr = res.setdefault(k, [0, 0, 0.0])
            rr[0] += first.sum(); rr[1] += Lsub * ng; rr[2] += dropped
            print(ci, k, "visited %.3f dropped-before-pixels %.3f median bound/true %.3f" % (first.sum() / (Lsub * ng), dropped, np.median(grp.sum(1) / total)))
    for k, rr in res.items():
        print(k, "visited %.4f  dropped %.3f" % (rr[0] / rr[1], rr[2] / n_cells))
main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
