"""Offline simulation (numpy + the oracle) of the filter's early-exit cascade on a few cells of a
bench configuration: how many 256-pixel groups each candidate visits before it is dropped, for
  (a) the v11 kernel: partial sums of the full fp32 dE00 + block bound of the unvisited groups;
  (b) per-pixel lower bound with the BLOCK's worst constants + the same block bound of the rest;
  (c) the same with S_C taken per pixel pair (chroma planes).
Not a test: a design tool (python tools/bound_sim.py [config] [n_cells])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_ffi  # noqa: E402
from test_filter_bound import rotation_bound, sqrt_ratio7  # noqa: E402

import bench  # noqa: E402


def walk_order(size):
    p = np.arange(size * size)
    y, x = p // size, p % size
    tiles_x = (size + 15) // 16
    key = (((y // 16) * tiles_x + x // 16) * 4 + ((y % 16) // 8) * 2 + (x % 16) // 8) * 64 + (y % 8) * 8 + x % 8
    return np.argsort(key, kind="stable")


WHATIF = os.environ.get("WHATIF", "")


def block_constants(cb, lb):
    """cb [nb][64][3] cell blocks, lb [L][nb][64][3] library blocks -> s_l, lam [L][nb]."""
    dev = 0.5 * (np.abs(cb[..., 0] - 50).max(-1)[None] + np.abs(lb[..., 0] - 50).max(-1))
    s_l = 1.0 + 0.015 * dev ** 2 / np.sqrt(20.0 + dev ** 2)
    cbar = 0.5 * (np.hypot(cb[..., 1], cb[..., 2]).max(-1)[None] + np.hypot(lb[..., 1], lb[..., 2]).max(-1))
    gc = (1.5 - 0.5 * sqrt_ratio7(cbar)) * cbar
    s_c = 1.0 + 0.045 * gc
    e_max = rotation_bound(np.broadcast_to(cb[None], lb.shape), lb)
    half_rt = sqrt_ratio7(gc) * np.sin(np.radians(60.0) * e_max) * 1.0001
    if WHATIF == "r0":
        half_rt = half_rt * 0.0
    if WHATIF == "rhalf":
        half_rt = half_rt * 0.5
    s_h = 1.0 + 0.015 * 1.5726 * gc
    a, c, b = 1.0 / s_c ** 2, 1.0 / s_h ** 2, half_rt / (s_c * s_h)
    lam = np.maximum(0.5 * (a + c) - np.sqrt(0.25 * (a - c) ** 2 + b ** 2), 0.0)
    rho = 1.0 - half_rt
    return s_l, lam, rho, 1.5 - 0.5 * sqrt_ratio7(cbar)


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "D"
    n_cells = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    w = bench.build_workload(config, 1)
    o = oracle_ffi.load()
    size, L = w["size"], w["L"]
    order = walk_order(size)
    lib = w["lib"].reshape(L, size * size, 3)[:, order].astype(np.float64)
    nb = size * size // 64
    lb = lib.reshape(L, nb, 64, 3)
    rng = np.random.default_rng(1)
    cells = w["cells_single"]
    pick = rng.choice(len(cells), n_cells, replace=False)
    tot = {k: 0 for k in ("a", "b", "c")}
    pre = 0
    surv = {k: 0 for k in ("b", "c")}
    for ci in pick:
        r = cells[ci]
        cell = w["img"][r["y"]:r["y"] + size, r["x"]:r["x"] + size].reshape(-1, 3)[order].astype(np.float64)
        cb = cell.reshape(nb, 64, 3)
        s_l, lam, rho, gmin = block_constants(cb, lb)
        d = np.abs(cb.sum(1)[None] - lb.sum(2))                                   # [L][nb][3]
        blk = np.sqrt((d[..., 0] / s_l) ** 2 + lam * ((gmin * d[..., 1]) ** 2 + d[..., 2] ** 2))       # block bound [L][nb]
        grp_rest = blk.reshape(L, nb // 4, 4).sum(2)
        rest = np.cumsum(grp_rest[:, ::-1], 1)[:, ::-1]                            # rest[:, j] = bound of groups >= j
        rest = np.concatenate([rest, np.zeros((L, 1))], 1)
        dpx = cell.reshape(1, nb, 64, 3) - lb                                      # [L][nb][64][3]
        pb = np.sqrt((dpx[..., 0] / s_l[..., None]) ** 2 + lam[..., None] * (dpx[..., 1] ** 2 + dpx[..., 2] ** 2))
        # (c) S_C per pixel pair: chroma planes; g(c) <= min(1.5 c, c + 6.4)
        cc = np.hypot(cb[..., 1], cb[..., 2])[None]
        lc = np.hypot(lb[..., 1], lb[..., 2])
        cm = 0.5 * (cc + lc)
        gpx = np.minimum(1.5 * cm, cm + 6.4)
        scpx = 1.0 + 0.045 * gpx
        pc = np.sqrt((dpx[..., 0] / s_l[..., None]) ** 2 + rho[..., None] * (dpx[..., 1] ** 2 + dpx[..., 2] ** 2) / scpx ** 2)
        pbg = pb.reshape(L, nb // 4, 256).sum(2)
        pcg = pc.reshape(L, nb // 4, 256).sum(2)
        # true scores of the most promising candidates (by the pixel bound) to find the limit
        lbsum = np.maximum(pbg.sum(1), pcg.sum(1))
        cand = np.argsort(lbsum)[:300]
        true = np.full(L, np.inf)
        cellf = np.broadcast_to(cell.astype(np.float32)[None], (len(cand), size * size, 3))
        tv = o.colour_diff_batch(2, cellf.reshape(-1, 3), lib[cand].astype(np.float32).reshape(-1, 3)).reshape(len(cand), -1)
        true[cand] = tv.sum(1)
        limit = np.sort(true)[K - 1] * 1.0003 + 8.0
        assert (lbsum[cand] <= true[cand] * (1 + 1e-9)).all()
        # full per-group true sums for (a): only for candidates not pruned by the prelude, sampled
        alive = rest[:, 0] <= limit
        pre += int((~alive).sum())
        ng = nb // 4

        def visits(grp):
            part = np.concatenate([np.zeros((grp.shape[0], 1)), np.cumsum(grp, 1)], 1)      # part[:, j] = groups < j
            stop = (part + rest[alive][:, :ng + 1]) > limit                                  # test before visiting group j
            first = np.where(stop.any(1), stop.argmax(1), ng + 1)
            return np.minimum(first, ng), int((first > ng).sum())
        vb, sb = visits(pbg[alive])
        vc, sc = visits(pcg[alive])
        tot["b"] += int(vb.sum()); tot["c"] += int(vc.sum()); surv["b"] += sb; surv["c"] += sc
        # (a) on a sample of the alive candidates
        ai = np.flatnonzero(alive)
        samp = ai if len(ai) <= 400 else rng.choice(ai, 400, replace=False)
        cellf = np.broadcast_to(cell.astype(np.float32)[None], (len(samp), size * size, 3))
        ta = o.colour_diff_batch(2, cellf.reshape(-1, 3), lib[samp].astype(np.float32).reshape(-1, 3)).reshape(len(samp), ng, 256).sum(2)
        part = np.concatenate([np.zeros((len(samp), 1)), np.cumsum(ta, 1)], 1)
        stop = (part + rest[samp][:, :ng + 1]) > limit
        first = np.where(stop.any(1), stop.argmax(1), ng)
        tot["a"] += first.sum() * len(ai) / len(samp)
        ratio_b = np.median(pbg[samp].sum(1) / ta.sum(1)); ratio_c = np.median(pcg[samp].sum(1) / ta.sum(1))
        ratio_blk = np.median(rest[samp, 0] / ta.sum(1))
        print("cell %d: limit %.0f alive %d/%d  bound/true medians: block %.2f px-block-const %.2f px-own-SC %.2f" % (
            ci, limit, alive.sum(), L, ratio_blk, ratio_b, ratio_c))
    total_groups = n_cells * L * (nb // 4)
    print("dropped by the prelude: %.3f of candidates" % (pre / (n_cells * L)))
    for k in ("a", "b", "c"):
        print("(%s) visited groups: %.3f of all" % (k, tot[k] / total_groups), "" if k == "a" else "survivors/cell %.1f" % (surv[k] / n_cells))


if __name__ == "__main__":
    sys.argv = sys.argv[:4]
    main()
