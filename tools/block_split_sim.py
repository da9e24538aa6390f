"""Offline simulation (numpy + the oracle): WHICH part of a finer block of the early-exit bound buys its gain.

tools/block_size_sim.py (round 5) found that 16-position blocks instead of 64 would walk 13 % fewer pixel groups on config D.  A finer
block changes two things at once: the SUMS of the triangle inequality (sum L, sum a, sum b over fewer positions: less cancellation) and the
CONSTANTS of the metric (S_L, S_C, lambda, g_min, the hue arcs: maxima over fewer pixels are tighter).  The sums are cheap to refine
(three subtractions and a root per sub-block), the constants are one stats_bound -- ~90 instructions, ten transcendental -- per block
pair.  This tool evaluates the bound with the two block sizes chosen separately ("c64/s16" = constants per 64 positions, sums per 16)
on a few cells x 600 candidates of config D, true fp64 group sums from the oracle, the kernel's visiting order.
Result (DESIGN.md section 4 "The finer block"): the sums buy nothing, the constants everything.
Not a test: a design tool (python tools/block_split_sim.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_ffi
from test_filter_bound import rotation_bound, sqrt_ratio7
from bound_sim import walk_order
import bench

def consts(cell, lib, bs):
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
    return s_l, lam, gmin      # [L][nb]

def bound(cell, lib, bs_const, bs_sum):
    L, P, _ = lib.shape
    s_l, lam, gmin = consts(cell, lib, bs_const)
    rep = bs_const // bs_sum
    s_l, lam, gmin = [np.repeat(x, rep, axis=1) for x in (s_l, lam, gmin)]
    nb = P // bs_sum
    d = np.abs(cell.reshape(nb, bs_sum, 3).sum(1)[None] - lib.reshape(L, nb, bs_sum, 3).sum(2))
    return np.sqrt((d[..., 0] / s_l) ** 2 + lam * ((gmin * d[..., 1]) ** 2 + d[..., 2] ** 2))

w = bench.build_workload("D", 1)
o = oracle_ffi.load()
size, L = w["size"], w["L"]; P = size*size; ng = P//256
order = walk_order(size)
lib = w["lib"].reshape(L, P, 3)[:, order].astype(np.float64)
rng = np.random.default_rng(1)
cells = w["cells_single"]
pick = rng.choice(len(cells), 3, replace=False)
variants = {"c64/s64": (64,64), "c64/s16": (64,16), "c64/s4": (64,4), "c256/s64": (256,64), "c256/s16": (256,16), "c16/s16": (16,16)}
dropped = {k:0 for k in variants}; walked={k:0.0 for k in variants}
for ci in pick:
    r = cells[ci]
    cell = w["img"][r["y"]:r["y"]+size, r["x"]:r["x"]+size].reshape(-1,3)[order].astype(np.float64)
    grp = {k: bound(cell, lib, *v).reshape(L, ng, -1).sum(2) for k,v in variants.items()}
    cand = np.argsort(grp["c16/s16"].sum(1))[:200]
    cellf = np.broadcast_to(cell.astype(np.float32)[None], (len(cand), P, 3))
    tv = o.colour_diff_batch(2, cellf.reshape(-1,3), lib[cand].astype(np.float32).reshape(-1,3)).reshape(len(cand),-1).sum(1)
    limit = np.sort(tv)[3]*1.0003+8.0
    samp = rng.choice(L, 600, replace=False)
    cellf = np.broadcast_to(cell.astype(np.float32)[None], (len(samp), P, 3))
    ta = o.colour_diff_batch(2, cellf.reshape(-1,3), lib[samp].astype(np.float32).reshape(-1,3)).reshape(len(samp), ng, 256).sum(2)
    for k in variants:
        g = grp[k][samp]
        assert (g <= ta*(1+1e-9)+1e-9).all(), k
        alive = g.sum(1) <= limit
        dropped[k] += int((~alive).sum())
        idx = np.argsort(-g, axis=1, kind="stable")
        gs, ts = np.take_along_axis(g, idx, 1), np.take_along_axis(ta, idx, 1)
        part = np.concatenate([np.zeros((len(samp),1)), np.cumsum(ts,1)],1)
        rest = np.concatenate([np.cumsum(gs[:,::-1],1)[:,::-1], np.zeros((len(samp),1))],1)
        stop = (part+rest) > limit
        first = np.where(stop.any(1), stop.argmax(1), ng)
        walked[k] += float(first[alive].sum())
n = 3*600
for k in variants:
    print("%-9s dropped %.3f walked %.4f" % (k, dropped[k]/n, walked[k]/(n*ng)))
