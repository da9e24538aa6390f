#!/usr/bin/env python3
"""Design tool (runs on the GPU box): how long a cell's candidate list has to be on photographic content, and whether the seeds
can tell beforehand.

bench.py --config PD (photographs at config D's scale) learns lists of 16 entries for every cell, and the fp64 re-score of 16 entries
per cell plus the looser pruning limit they give the filter are 40 % of its step.  This script takes the device's full fp64 score
rows (mm_score_cells with k = L: 4096 x 5000), the final grid of a generate, and computes per cell
  need   = entries the verifying resolve needs to prove the final pick: the pick's total (base + penalty) must lie below the list's last base score;
  seeds  = the M images closest in mean colour (what seed_kernel picks, approximated on the host), with their exact scores;
  rule   = k_c = clamp(#{seeds within T of the best seed} + c0, 4, Kmax)
and reports, for a grid of (T, c0, Kmax): the mean list length, the share of cells at Kmax, and the share whose rule length is below
their need (those would be deepened).  usage: python tools/list_length_sim.py [config]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    from mosaicmagnifique_amd.engine import Engine
    from mosaicmagnifique_amd.grid import CELL_DTYPE
    name = sys.argv[1] if len(sys.argv) > 1 else "PD"
    w = bench.build_workload(name, 1)
    e = Engine(0)
    bench.stage(e, w)
    cells = np.ascontiguousarray(w["cells"], dtype=CELL_DTYPE)
    L, t, (rng, add) = w["L"], w["t"], w["rep"]
    best, score = e.generate(cells, w["grid_w"], w["grid_h"], t, rng, add)
    idx, sc = e.score_cells(cells, t, L)                       # full rows (k >= L: in index order)
    order = np.lexsort((idx, sc), axis=1)                      # ascending by (score, index), as the lists are
    idx, sc = np.take_along_axis(idx, order, 1), np.take_along_axis(sc, order, 1)
    n = len(cells)
    pick = best[cells["grid_y"], cells["grid_x"]]
    total = score[cells["grid_y"], cells["grid_x"]]
    rank = np.array([int(np.nonzero(idx[c] == pick[c])[0][0]) for c in range(n)])
    # entries needed: the first list position whose base score exceeds the pick's total (+ a tie band), plus one
    need = np.array([int(np.searchsorted(sc[c], total[c] * (1 + 1e-9), side="right")) + 1 for c in range(n)])
    print("config %s: %d cells, %d images; rank of the final pick: mean %.2f, max %d; entries needed: mean %.2f, median %d, p90 %d, p99 %d, max %d" % (
        name, n, L, rank.mean(), rank.max(), need.mean(), np.median(need), np.percentile(need, 90), np.percentile(need, 99), need.max()))
    for k in (4, 8, 16, 24, 42):
        print("  fixed lists of %2d: %.1f %% of the cells would be deepened" % (k, 100.0 * (need > k).mean()))
    # seeds by mean colour
    img, lib = bench.host_arrays(e, w)
    size = w["size"]
    lib_mean = lib.reshape(L, -1, 3).mean(1)
    cell_mean = np.stack([img[max(r["y"], 0):r["y"] + size, max(r["x"], 0):r["x"] + size].reshape(-1, 3).mean(0) for r in cells])
    M = 30
    d2 = ((cell_mean[:, None, :] - lib_mean[None, :, :]) ** 2).sum(-1)          # [n][L]
    seeds = np.argsort(d2, axis=1)[:, :M]
    # exact scores of the seeds: position of each seed in the cell's sorted row
    inv = np.empty((n, L), dtype=np.int32)
    rows = np.arange(n)[:, None]
    inv[rows, idx] = np.arange(L, dtype=np.int32)[None, :]
    seed_s = np.sort(sc[rows, inv[rows, seeds]], axis=1)                         # [n][M] ascending
    print("  seeds (M = %d by mean colour): the best seed is the cell's true best in %.1f %% of the cells; true rank of the best seed: mean %.2f" % (
        M, 100.0 * (seed_s[:, 0] == sc[:, 0]).mean(), np.array([np.searchsorted(sc[c], seed_s[c, 0]) for c in range(n)]).mean()))
    print("  rule k_c = clamp(#seeds within T of the best seed + c0, 4, Kmax):  T, c0, Kmax -> mean k, share at Kmax, share deepened (k_c < need), pruning limit / fixed-16 limit (median)")
    lim16 = seed_s[:, 15]
    for kmax in (16, 24):
        for T in (250, 500, 1000, 1500, 2500, 4000):
            for c0 in (1, 2, 4):
                cnt = (seed_s <= seed_s[:, :1] + T * abs(add) / 500.0).sum(1)
                kc = np.clip(cnt + c0, 4, kmax)
                lim = seed_s[np.arange(n), kc - 1]
                print("    T %5d c0 %d Kmax %2d -> mean k %5.2f, at Kmax %5.1f %%, deepened %5.2f %%, limit ratio %.4f" % (
                    T, c0, kmax, kc.mean(), 100.0 * (kc == kmax).mean(), 100.0 * (kc < need).mean(), float(np.median(lim / lim16))))
    # a neighbourhood predictor: how many cells of the window of calculateRepeats (rows above within +-R columns, the row's cells to the
    # left) have the same best seed -- or share one of their best `top` seeds -- as this cell: collisions are what lengthen a list
    seed_order = np.argsort(sc[rows, inv[rows, seeds]], axis=1)
    top_ids = np.take_along_axis(seeds, seed_order, 1)                               # [n][M] seed ids by ascending exact score
    gh, gw = w["grid_h"], w["grid_w"]
    cell_at = -np.ones((gh, gw), dtype=np.int64)
    cell_at[cells["grid_y"], cells["grid_x"]] = np.arange(n)
    print("  rule k_c = clamp(a x (window neighbours sharing one of the best `top` seeds) + c0, 4, Kmax): top, a, c0, Kmax -> mean k, at Kmax, deepened")
    for top in (1, 2, 3):
        share = np.zeros(n, dtype=np.int64)
        mine = [set(top_ids[c, :top].tolist()) for c in range(n)]
        for c in range(n):
            x, y = int(cells["grid_x"][c]), int(cells["grid_y"][c])
            nb = []
            for yy in range(max(y - rng, 0), y):
                nb += cell_at[yy, max(x - rng, 0):min(x + rng, gw - 1) + 1].tolist()
            nb += cell_at[y, max(x - rng, 0):x].tolist()
            share[c] = sum(1 for m in nb if m >= 0 and mine[m] & mine[c])
        print("    top %d: neighbours sharing: mean %.2f, max %d; correlation with need %.3f" % (top, share.mean(), share.max(), float(np.corrcoef(share, need)[0, 1])))
        for kmax in (16,):
            for a in (0.5, 1.0, 1.5):
                for c0 in (2, 3, 4):
                    kc = np.clip(np.ceil(a * share).astype(np.int64) + c0, 4, kmax)
                    print("    top %d a %.1f c0 %d Kmax %2d -> mean k %5.2f, at Kmax %5.1f %%, deepened %5.2f %%" % (
                        top, a, c0, kmax, kc.mean(), 100.0 * (kc == kmax).mean(), 100.0 * (kc < need).mean()))
    # ... and the same from the PROXY alone (the library images closest in mean colour, before anything is scored): what a host could
    # know before the first scoring pass, to split the cells into a short-list and a long-list batch
    print("  two classes from the proxy order: class LONG (lists of 16) iff >= thr window neighbours share one of their `top` closest-mean images; else lists of `short`")
    for top in (1, 2, 3, 4):
        mine = [set(seeds[c, :top].tolist()) for c in range(n)]
        share = np.zeros(n, dtype=np.int64)
        for c in range(n):
            x, y = int(cells["grid_x"][c]), int(cells["grid_y"][c])
            nb = []
            for yy in range(max(y - rng, 0), y):
                nb += cell_at[yy, max(x - rng, 0):min(x + rng, gw - 1) + 1].tolist()
            nb += cell_at[y, max(x - rng, 0):x].tolist()
            share[c] = sum(1 for m in nb if m >= 0 and mine[m] & mine[c])
        print("    top %d (proxy): correlation with need %.3f" % (top, float(np.corrcoef(share, need)[0, 1])))
        for short in (4, 6):
            for thr in (1, 2, 3, 5):
                kc = np.where(share >= thr, 16, short)
                print("    top %d short %d thr %d -> LONG class %5.1f %% of the cells, mean k %5.2f, deepened %5.2f %%" % (
                    top, short, thr, 100.0 * (kc == 16).mean(), kc.mean(), 100.0 * (kc < need).mean()))
    e.close()


if __name__ == "__main__":
    main()
