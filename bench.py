#!/usr/bin/env python3
"""Headline benchmark: cells matched / second on the photomosaic best-fit path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config D|A|B|C|E|E8]
                    [--scaling weak|strong] [--mcs FILE.mcs] [--mil FILE.mil]

A "step" is one full pass of the hot path (generateBestFits for one size step)
over the whole cell batch, with library, masks and target already resident in
HBM.  Default workload = BASELINE.json config D, the one the north-star target is
quoted on: 4096x4096 target, 64x64 square cells (4096 cells), 5000-image library,
CIEDE2000, repeat range 4 (+500).

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL):
  * config D (default): WEAK scaling -- every rank scores one config-D band of cells
    (the target is the config-D image repeated per band, so the job is a
    4096 x 4096*N mosaic), one all-gather of the candidate lists, identical
    resolve on every rank;
  * config E (BASELINE config 5: 8192x8192 target, 48x48 cells -> 171 x 171 = 29 241
    cells with the last row / column clipped to 32 px, 10 000 images, CIEDE2000):
    STRONG scaling -- the fixed cell list is cut into ceil(29 241 / N) contiguous
    shards, one all-gather, identical resolve.  --scaling overrides either default.
Communicator set-up and staging happen before the timed region and are reported
(`stage_once_s`, `comm_init_s`).

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel
(filter_kernel): for CIEDE2000 it is bound by VALU issue, so `achieved` is its VALU
wave-instruction rate in fma-equivalent TFLOP/s (128 flop per wave-instruction)
against the fp32 vector peak; the instruction count comes from the rocprofv3 SQ pass
recorded in profiles/pmc_traffic.json for exactly these kernel sources and this
configuration (checked by a hash of the sources and by the launch's own always-on
visited-groups counter).  `roofline.hbm` is the measured fabric-side traffic of the
same launch against HBM peak; the bytes a naive stream would move are
kept as `effective_algorithmic_gbs` and are NOT a fraction of anything.
`cpu_baseline` is the CPU oracle (a port of the reference's single-threaded
generator, early exit on) timed on a bounded prefix of the same workload, rank 0
at N=1 only.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction x 128 flop
VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 2.0
FORM_VALU_PER_PAIR = 86.0      # VALU instructions of ciede2000_fast per pixel pair in the ISA of filter_kernel (344 per four pairs)

CONFIGS = {
    # name: (rows, cols, cell, shape, L, diff type, repeat range, addition, cells of the 1-thread CPU sample)
    "A": (512, 512, 32, "square", 100, 0, 0, 0, 256),
    "B": (2048, 2048, 64, "square", 1000, 0, 0, 0, 256),
    "C": (2048, 2048, 64, "hexagon", 1000, 2, 0, 0, 48),
    "D": (4096, 4096, 64, "square", 5000, 2, 4, 500, 8),
    "E": (8192, 8192, 48, "square", 10000, 2, 0, 0, 4),      # BASELINE config 5, the whole grid
    "E8": (8192, 8192, 48, "square", 10000, 2, 0, 0, 4),     # ... one rank's shard of it (first ceil(29 241 / 8) cells)
    # the reference GUI's defaults -- cell size 128 (CellShape.h:36), detail 100 %, the first colour difference of the combo box
    # (MainWindow.cpp:60,150,157) -- on a 4096 x 4096 target with a 1000-image library, and the same shape as CIEDE2000
    "F": (4096, 4096, 128, "square", 1000, 0, 0, 0, 16),
    "G": (4096, 4096, 128, "square", 1000, 2, 0, 0, 2),
    # config D against a library that stays in the Infinity Cache (500 images: 33 MB interleaved): what the filter kernel's
    # instruction rate is when the fabric is out of the way
    # the reference harness's default line as CIEDE2000 (Benchmark_Generator.h: 5000 x 4000, cell 128 at detail 20 % = 25 px detail
    # cells, ~200 images), at the detail size directly: 1280 cells of 25 x 25 against 214 images
    "H": (800, 1000, 25, "square", 214, 2, 0, 0, 64),
    # 80 x 80 cells (25 groups of 256 pixels: the class between 64 x 64 and the GUI's 128 x 128), CIEDE2000
    "I": (4000, 4000, 80, "square", 1000, 2, 0, 0, 4),
    # config D's parameters on PHOTOGRAPHIC content (tests/golden/make_photo_fixture.py: one of the reference's SampleImages as the
    # target, 1024 tiles of 64 x 64 cut from the other four as the library): what the early-exit bound does off the synthetic patches
    "P": (1363, 2048, 64, "square", 1024, 2, 4, 500, 16),
    # ... and at config D's own scale (tests/golden/make_photo_fixture_pd.py): one photograph as a 4096 x 4096 target, 5000 tiles of 64 x 64 cut
    # from the other four at two zoom levels, CIEDE2000, repeat range 4 (+500) -- config D's job shape on the content the tool is for
    "PD": (4096, 4096, 64, "square", 5000, 2, 4, 500, 8),
    "D500": (4096, 4096, 64, "square", 500, 2, 4, 500, 8),
    "D3800": (4096, 4096, 64, "square", 3800, 2, 4, 500, 8),     # ... and one whose interleaved copy (249 MB) just fits the 256 MB Infinity Cache
    # reduced shapes of D and E for the rank rehearsals of tests/test_gpu_bench_ranks.py (not bench lines)
    "Ds": (1024, 1024, 64, "square", 400, 2, 4, 500, 8),
    "Es": (1000, 1000, 48, "square", 500, 2, 0, 0, 4),       # 21 x 21 cells, last row / column clipped to 40 px
}
TYPE_NAMES = {0: "RGB-Euclidean", 1: "CIE76", 2: "CIEDE2000"}


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the engine's kernel sources: ties a PMC record to the CODE it measured -- comments and
    white space are taken out first (a record stays valid through an edit that changes no token of the code)."""
    import re
    from mosaicmagnifique_amd import _build
    h = hashlib.sha256()
    for path in sorted(_build.SOURCES + [p for p in _build.HEADERS if os.sep + "csrc" + os.sep in p]):
        text = open(path, "r", encoding="utf-8", errors="replace").read()
        # (string literals of these sources hold neither "//" nor "/*" except in asm templates, which hold none either)
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode("utf-8"))
    return h.hexdigest()[:16]


def result_digest(best, score):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(best, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(np.nan_to_num(score, nan=-1.0), dtype=np.float64).tobytes())
    return h.hexdigest()


def build_workload(name, world, scaling="weak", mcs=None, mil=None, bands=0, host_preprocess=False):
    import cases
    from mosaicmagnifique_amd import GridUtility as G
    from mosaicmagnifique_amd import synth
    rows, cols, size, shape_name, L, t, rng, add, cpu_cells = CONFIGS[name]
    if mcs:                                   # the project's own cell shape file, resized to the config's cell size
        from mosaicmagnifique_amd.cellshape import CellShape
        shape = CellShape.loadFromFile(mcs).resized(size)
        shape_name = "%s (%s)" % (shape.name or "mcs", os.path.basename(mcs))
    else:
        shape = cases.hexagon_shape() if shape_name == "hexagon" else cases.square_shape(size)
    group = cases.group_of(shape)
    if mil:                                   # the project's own image library file
        from mosaicmagnifique_amd import imageutil
        from mosaicmagnifique_amd.imagelibrary import ImageLibrary
        library = ImageLibrary.loadFromFile(mil)
        lib_u8 = library.getImages()
        if library.getImageSize() != size:
            lib_u8 = np.stack([imageutil.resize_area_u8(im, size, size) for im in lib_u8])
        L = len(lib_u8)
    elif name in ("P", "PD"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        if name == "P":
            import make_photo_fixture
            photo_u8, lib_u8 = make_photo_fixture.load()
        else:
            import make_photo_fixture_pd
            photo_u8, lib_u8 = make_photo_fixture_pd.load()
        assert photo_u8.shape == (rows, cols, 3) and lib_u8.shape == (L, size, size, 3)
    else:
        lib_u8 = synth.library_u8(L, size)
    state1 = G.getGridState(group, rows, cols)[0]
    cells1 = G.cell_records(group, state1, rows, cols)
    band_rows_px = rows
    if name == "E8":                          # one rank's shard: the band of the target holding its cells
        n_shard = -(-len(cells1) // 8)
        cells1 = cells1[:n_shard]
        band_rows_px = min(rows, (int(cells1["y"].max()) // size + 2) * size)
    img_u8 = photo_u8 if name in ("P", "PD") else synth.target_u8(band_rows_px, cols, lattice=64)
    # convert on the device (mm_set_*_u8) instead of ~1 min of numpy per rank -- unless the caller wants the arrays the committed CPU
    # golden was computed on (numpy's float64 Lab conversion; the device's differs from it in the last place of single f32 values)
    big = band_rows_px * cols > 4096 * 4096 and not host_preprocess
    w = dict(name=name, shape=shape, shape_name=shape_name, t=t, rep=(rng, add), L=L, size=size,
             rows=band_rows_px, cols=cols, cpu_cells=cpu_cells, state_single=state1,
             device_preprocess=big, img_u8=img_u8, lib_u8=lib_u8, scaling=scaling, custom_inputs=bool(mcs or mil))
    if not big:
        w["img"] = synth.preprocess(img_u8, t)
        w["lib"] = synth.preprocess(lib_u8, t)
    bands = bands or world
    if bands == 1 or scaling == "strong":
        cells, grid_w, grid_h = cells1, state1.shape[1], state1.shape[0]
    else:
        world = bands
        # weak scaling: `world` bands, each the single-GPU image; band b's cells read the
        # same target pixels (y unchanged) but sit `b * band_rows` grid rows lower.
        if shape_name != "square":
            raise SystemExit("weak-scaled bands are defined for square cells")
        band_rows = state1.shape[0] - G.PAD_GRID
        parts = []
        for b in range(world):
            c = cells1.copy()
            c["grid_y"] += b * band_rows
            parts.append(c)
        cells = np.concatenate(parts)
        grid_w, grid_h = state1.shape[1], band_rows * world + G.PAD_GRID
    w.update(cells=cells, cells_single=cells1, grid_w=grid_w, grid_h=grid_h)
    return w


def stage(engine, w):
    """Library, masks and target to HBM (once, before the timed region)."""
    if w["device_preprocess"]:
        engine.set_library_u8(w["lib_u8"], w["t"])
        engine.set_target_u8(w["img_u8"], w["t"])
    else:
        engine.set_library(w["lib"])
        engine.set_target(w["img"])
    engine.set_masks(w["shape"].masks4())


def host_arrays(engine, w):
    """The f32 arrays the engine holds, for the CPU oracle (identical inputs on both sides)."""
    if "img" not in w:
        w["img"] = engine.get_target(w["rows"], w["cols"])
        w["lib"] = engine.get_library()
    return w["img"], w["lib"]


def algorithmic_bytes(w, cells):
    """per valid cell L*Pa*12 + Pa*12 + Pa, Pa = active (mask AND bound) pixels."""
    masks = w["shape"].masks4() != 0
    size = w["size"]
    full = masks.reshape(4, -1).sum(1)
    total_pa = 0
    integ = [np.pad(m.astype(np.int64).cumsum(0).cumsum(1), ((1, 0), (1, 0))) for m in masks]
    for mid in range(4):
        sel = cells[cells["mask_id"] == mid]
        if not len(sel):
            continue
        whole = (sel["row_min"] == 0) & (sel["row_max"] == size) & (sel["col_min"] == 0) & (sel["col_max"] == size)
        total_pa += int(whole.sum()) * int(full[mid])
        c = sel[~whole]
        I = integ[mid]
        total_pa += int((I[c["row_max"], c["col_max"]] - I[c["row_min"], c["col_max"]]
                         - I[c["row_max"], c["col_min"]] + I[c["row_min"], c["col_min"]]).sum())
    return total_pa * (w["L"] * 12 + 13), total_pa


def host_cores(cap=16):
    """Threads for the all-cores leg: the affinity mask, cut by the cgroup CPU quota when there
    is one, and by `cap` (a one-GPU box's CPU share is 16 cores whatever the mask says)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline(w, img, lib):
    """The oracle (port of CPUPhotomosaicGenerator, 1 thread, early exit) on the
    first `cpu_cells` valid cells in raster order of the same inputs."""
    import oracle_ffi
    o = oracle_ffi.load()
    n = w["cpu_cells"]
    t0 = time.perf_counter()
    grid, scores, done = o.generate_best_fits(
        oracle_ffi.shape_of(w["shape"]), img, lib, w["shape"].masks4(), w["t"],
        w["rep"][0], w["rep"][1], max_cells=n, state=w["state_single"])
    dt = time.perf_counter() - t0
    return grid, scores, done, dt


def parity_sample(w, my_cells):
    """Which cells the all-cores parity leg checks.  With a repeat penalty: the first two full grid rows in raster
    order (the window's rows-above half is exercised, and equality on a prefix is equality with the sequential run).
    Without one the cells are independent: 64 of them drawn at random (seeded) from the whole workload -- for shapes
    with clipped border cells (config C's hexagons) 64 unclipped + 16 clipped ones that hold active pixels."""
    rep, size = w["rep"], w["size"]
    if rep[0] > 0:
        rows = np.unique(my_cells["grid_y"])
        if w.get("golden_ok") and len(rows) > 2 * rep[0] + 4:
            # the whole grid equals the committed CPU golden (checked first), so the induction over the raster order has its base
            # anywhere: two full grid rows at a seeded INTERIOR offset, the full rows-above window of calculateRepeats in front of them
            first = int(np.random.default_rng(0xC0FFEE + w["L"]).integers(rep[0] + 1, len(rows) - 2))
            pick = my_cells[(my_cells["grid_y"] == rows[first]) | (my_cells["grid_y"] == rows[first + 1])]
            return pick[:256], "%%d cells: two full grid rows at a seeded interior offset (grid rows %d, %d: the whole %d-rows-above window of calculateRepeats is in play; the earlier rows equal the CPU golden)" % (
                int(rows[first]), int(rows[first + 1]), rep[0])
        row_cells = int((my_cells["grid_y"] <= my_cells["grid_y"][0] + 1).sum())
        return my_cells[:min(len(my_cells), max(row_cells, 16), 256)], "first %d valid cells in raster order (2 full grid rows: the repeat window's rows-above half is exercised)"
    rng = np.random.default_rng(0xC0FFEE)
    whole = (my_cells["row_min"] == 0) & (my_cells["row_max"] == size) & (my_cells["col_min"] == 0) & (my_cells["col_max"] == size)
    masks = w["shape"].masks4() != 0
    active = np.array([masks[c["mask_id"]][c["row_min"]:c["row_max"], c["col_min"]:c["col_max"]].any() for c in my_cells[~whole]], dtype=bool)
    inner, clipped = np.flatnonzero(whole), np.flatnonzero(~whole)[active] if (~whole).any() else np.zeros(0, dtype=np.int64)
    pick = np.concatenate([rng.choice(inner, size=min(64, len(inner)), replace=False),
                           rng.choice(clipped, size=min(16, len(clipped)), replace=False)]).astype(np.int64)
    pick.sort()
    return my_cells[pick], "%%d cells drawn at random from the whole workload (%d unclipped + %d clipped with active pixels, seed 0xC0FFEE; no repeat penalty: cells are independent)" % (
        min(64, len(inner)), min(16, len(clipped)))


def golden_parity(w, img, lib, best, score, bands=1, only_cells=None):
    """The WHOLE grid against the committed CPU golden of the configuration (tests/golden/grid_<config>.npz, the oracle's run over
    every cell: tests/golden/make_grids.py) -- when the arrays the engine holds are the ones the golden was computed on
    (sha256 of both; arrays converted on the device are read back and hashed like any others).  None if there is no golden for
    this workload.  One GPU and N strong-scaled shards end with the golden's own grid; of N weak-scaled bands (the configuration's
    image repeated down the mosaic) band 0 IS the golden's job -- the window of calculateRepeats looks up and left only
    (CPUPhotomosaicGenerator.cpp:185-225), so nothing below reaches it -- and without a repeat penalty every band is."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    try:
        import make_grids
    except ImportError:
        return None
    if w["name"] not in make_grids.CONFIGS or w["name"] == "d" or w.get("custom_inputs"):
        return None
    got = make_grids.load(w["name"])
    if got is None:
        return None
    want, want_score, meta = got
    if make_grids.sha(img) != meta["img_sha256"] or make_grids.sha(lib) != meta["lib_sha256"]:
        return {"cells": 0, "indices_equal": None, "note": "the golden was computed on other arrays (numpy rounds differently on this host?): not compared"}
    gh1 = w["state_single"].shape[0]
    band_rows = gh1 - 2                        # GridUtility.PAD_GRID
    n_bands = bands if len(w["cells"]) != len(w["cells_single"]) else 1
    compared = n_bands if w["rep"][0] == 0 else 1
    equal, differing, worst, cells = True, 0, 0.0, 0
    valid = want >= 0
    if only_cells is not None:                 # a shard of the grid (config E8): the golden's entries of exactly these cells
        valid = np.zeros_like(valid)
        valid[only_cells["grid_y"], only_cells["grid_x"]] = True
        n_bands = 2                            # (compare the valid cells only: the rest of the grid was not computed)
        compared = 1
    for b in range(compared):
        mine = np.where(best < w["L"], best, -1)[b * band_rows:b * band_rows + gh1]
        sc = score[b * band_rows:b * band_rows + gh1]
        if b > 0 or n_bands > 1:               # the two padding rows at a band seam belong to the neighbouring band: valid cells only
            same = np.array_equal(mine[valid], want[valid])
        else:
            same = np.array_equal(mine, want)
        equal = equal and same
        differing += int((mine[valid] != want[valid]).sum())
        rel = np.abs(sc[valid] - want_score[valid]) / np.maximum(np.abs(want_score[valid]), 1e-300)
        rel = rel[want_score[valid] != 0]
        worst = max(worst, float(rel.max()) if len(rel) else 0.0)
        cells += int(valid.sum())
    out = {"cells": cells, "indices_equal": bool(equal), "cells_differing": differing, "max_rel_score_err": worst,
           "golden": "tests/golden/grid_%s.npz (oracle/oracle.c over the whole grid, %s mode, %.0f s on %d host threads)" % (
               w["name"], meta["mode"], meta["seconds"], meta["threads"])}
    if n_bands > 1 and only_cells is None:
        out["bands_compared"] = ("all %d (no repeat penalty: every band is the golden's job)" % n_bands) if compared == n_bands else \
            "band 0 of %d (the repeat window looks up and left only; later bands: parity_on_cpu_sample)" % n_bands
    return out


def cpu_parity_rows(w, img, lib, best, recs):
    """index parity where the repeat window is exercised -- the first `n_cells` valid
    cells in raster order (config D: 2 full rows), each evaluated by the oracle exactly as the
    reference's raster loop reaches it with the earlier cells at the engine's picks; equality on
    the whole prefix is equality with the sequential run (induction over the raster order,
    oracle_ffi.picks_given_grid).  One thread per host core: also the all-cores CPU rate."""
    import oracle_ffi
    o = oracle_ffi.load()
    pos = np.stack([recs["grid_y"], recs["grid_x"]], axis=1)
    grid = np.where(best < w["L"], best, -1)[:w["state_single"].shape[0]]
    cores = host_cores()
    t0 = time.perf_counter()
    picks, scores = o.picks_given_grid(oracle_ffi.shape_of(w["shape"]), img, lib, w["shape"].masks4(), w["t"],
                                       w["rep"][0], w["rep"][1], grid, pos, threads=cores)
    dt = time.perf_counter() - t0
    return pos, picks, scores, cores, dt


def floor_legs(engine, step, n_cells, value):
    """What the headline owes to the content: the same step with the early exit off (`prune` 0: every candidate's every
    pixel through the fp32 filter) and on the fp64-only path (`filter` 0: the reference's arithmetic for every pair) --
    both produce the identical grid (tests).  Outside the timed region; up to three steps each, one if a step takes > 4 s."""
    import torch
    out = {}
    for key, option in (("floor_unpruned", "prune"), ("floor_fp64_only", "filter")):
        engine.set_option(option, 0)
        try:
            times = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
                if times[-1] > 4.0:
                    break
            dt = min(times)
        finally:
            engine.set_option(option, 1)
        out[key] = {"value": n_cells / dt, "unit": "cells/s", "ms_per_step": dt * 1e3, "steps": len(times),
                    "headline_over_floor": value / (n_cells / dt),
                    "what": "early exit off (option prune=0): every pixel of every candidate through the fp32 filter" if option == "prune"
                            else "fp64 only (option filter=0): every pixel pair in fp64 (the algebraic form of the parallel sums), no filter tier"}
    return out


def config_legs(device, opts, names, steps=4):
    """The other BASELINE configurations (and photographs at config D's scale) in the SAME run, the way the reference's harness sweeps
    its configurations in one process (Benchmark/Benchmark_Generator.h:125-183): per configuration a fresh engine, staging, one warm-up
    step, `steps` timed steps (wall clock between two device synchronisations), the dominant kernel's launch time from HIP events on
    the engine's stream, and the WHOLE grid of the last step against the committed CPU golden.  "E8" is config E's first
    ceil(29 241 / 8) cells -- one rank's shard of BASELINE config 5 -- on the engine that holds config E."""
    import torch
    from mosaicmagnifique_amd.engine import Engine
    from mosaicmagnifique_amd.grid import CELL_DTYPE
    out = {}
    todo = [n for n in names if n != "E8"]
    for name in todo:
        t_all = time.perf_counter()
        w = build_workload(name, 1, host_preprocess=True)      # (the golden's own input arrays: config E costs ~10 s of numpy)
        e = Engine(device)
        try:
            for item in opts:
                k, v = item.split("=")
                e.set_option(k, int(v))
            stage(e, w)
            runs = [(name, np.ascontiguousarray(w["cells"], dtype=CELL_DTYPE))]
            if name == "E" and "E8" in names:
                runs.append(("E8", runs[0][1][:-(-len(runs[0][1]) // 8)]))
            img = lib = None
            for label, cells in runs:
                gw, gh = w["grid_w"], w["grid_h"]
                arrays = (np.empty((gh, gw), dtype=np.int64), np.empty((gh, gw), dtype=np.float64))
                run = lambda: e.generate(cells, gw, gh, w["t"], *w["rep"], out=arrays)
                for _ in range(5 if w["rep"][0] > 0 else 1):      # (with a repeat penalty the engine tries up to three list lengths over its first generates, the base one twice)
                    run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    best, score = run()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                tm = e.timing()
                # the dominant kernel's time per STEP: its one launch, or -- photographs: deepening rounds -- the sum of its launches
                kernel_ms, kernel_launches = tm["score_ms"], tm["score_launches"]
                if kernel_ms == 0.0:                # a small job runs without phase events: one more step with them on
                    e.set_option("phase_timing", 1)
                    run()
                    tp = e.timing()
                    kernel_ms, kernel_launches = tp["score_ms"], tp["score_launches"]
                    e.set_option("phase_timing", -1)
                if img is None:
                    img, lib = host_arrays(e, w)
                gold = golden_parity(w, img, lib, best, score, only_cells=cells if label == "E8" else None)
                alg_bytes, total_pa = algorithmic_bytes(w, cells)
                out[label] = {
                    "workload": "%dx%d target, %dx%d %s cells, %d images, %s, repeat %d (+%d)%s" % (
                        w["cols"], w["rows"], w["size"], w["size"], w["shape_name"], w["L"], TYPE_NAMES[w["t"]], w["rep"][0], w["rep"][1],
                        ", first 1/8 of the cells (one rank's shard of BASELINE config 5)" if label == "E8" else ""),
                    "cells": int(len(cells)), "cells_per_s": len(cells) / dt, "ms_per_step": dt * 1e3, "steps": steps,
                    "kernel_ms": kernel_ms, "kernel_launches": kernel_launches, "kernel": "filter_kernel" if w["t"] == 2 else "filter_lds_kernel",
                    "visited_fraction_of_groups": tm["visited_groups"] / tm["total_groups"] if tm["total_groups"] else None,
                    "host_syncs": tm["host_syncs"], "resolve_sweeps": tm["resolve_sweeps"], "deepened_cells": tm["deepened_cells"],
                    "section_8d_algorithmic_gbs": alg_bytes / (kernel_ms / 1e3) / 1e9 if kernel_ms > 0 and kernel_launches == 1 else None,
                    "parity_on_golden": ({k: gold[k] for k in ("cells", "indices_equal", "cells_differing", "max_rel_score_err") if k in gold}
                                         if gold else None),
                    "band_alarms": tm["band_alarms"], "voided_launches": tm["voided_launches"],
                    "spec_rounds": tm["spec_rounds"], "filter_tile": tm["filter_tile"],
                }
            for label, _ in runs:
                out[label]["wall_s_incl_staging_and_golden"] = round(time.perf_counter() - t_all, 1)
        finally:
            e.close()
    return out


def first_call_leg(w, device, opts):
    """What generateBestFits() of a FRESH generator costs in a warm process (the reference builds one per run,
    MainWindow.cpp:584-607; its harness times exactly that, Benchmark_Generator.h:83-104): a new engine, the u8 inputs
    over PCIe, pre-processing, planes / statistics, the generate, the result on the host -- and the same once more with
    another fresh engine (which adopts what the first one parked when it was destroyed).  Wall ms + the engine's stage clocks."""
    from mosaicmagnifique_amd.engine import Engine
    from mosaicmagnifique_amd.grid import CELL_DTYPE
    cells = np.ascontiguousarray(w["cells"], dtype=CELL_DTYPE)
    masks = w["shape"].masks4()
    out = {}
    for label in ("first", "again"):
        t0 = time.perf_counter()
        e = Engine(device)
        for item in opts:
            name, val = item.split("=")
            e.set_option(name, int(val))
        e.set_library_u8(w["lib_u8"], w["t"])
        e.set_target_u8(w["img_u8"], w["t"])
        e.set_masks(masks)
        t1 = time.perf_counter()
        e.generate(cells, w["grid_w"], w["grid_h"], w["t"], *w["rep"])
        t2 = time.perf_counter()
        clocks = e.stage_timing()
        e.close()
        out[label] = {"first_call_ms": (t2 - t0) * 1e3, "stage_ms": (t1 - t0) * 1e3, "generate_ms": (t2 - t1) * 1e3,
                      "engine_clocks": {k: round(v, 3) if isinstance(v, float) else v for k, v in clocks.items()}}
    out["first_call_ms"] = out["first"]["first_call_ms"]
    out["what"] = ("fresh engine -> mm_set_library_u8 + mm_set_target_u8 + mm_set_masks + mm_generate -> grid on the host, wall ms in a "
                   "warm process (HIP context and code objects loaded); `again`: a second fresh engine after the first was destroyed")
    return out


def pmc_record(w, opts, tile=None):
    """rocprofv3 PMC passes of this same command (tools/profile.sh), summarised per launch of the dominant
    kernel by profiles/summarise_pmc.py and copied into profiles/ by profiles/collect.py.  Only a record measured on exactly these kernel sources, this
    configuration and these engine options is used; anything else is reported as absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_traffic.json missing"
    entry = json.load(open(path)).get(w["name"])
    if not entry:
        return None, "no PMC record for config %s" % w["name"]
    if entry.get("kernel_src_sha16") != kernel_source_hash():
        return None, "PMC record is for kernel sources %s, this build is %s" % (entry.get("kernel_src_sha16"), kernel_source_hash())
    if sorted(entry.get("options", [])) != sorted(opts):
        return None, "PMC record was taken with options %s" % entry.get("options")
    if tile is not None and entry.get("filter_tile") not in (None, tile):
        # (the item size can follow the engine's call history -- option "item_auto" -- so the record names the one it ran with)
        return None, "PMC record was taken with %s-candidate items, this run used %s" % (entry.get("filter_tile"), tile)
    return entry, "profiles/pmc_traffic.json (rocprofv3 --pmc, separate passes: FETCH_SIZE x2 + WRITE_SIZE; SQ_INSTS_VALU)"


def stage_bench(args, emit):
    """--stage build | preprocess: the HBM-bound byte kernels either side of the hot path, on
    the configuration's own sizes, priced against HBM peak with the bytes an ideal pass moves.
      build       mm_build_photomosaic on the finished best-fit grid: owner map (atomicMax, 4 B written per covered
                  pixel) + compose (4 B owner read + 3 B library read + 4 B BGRA written + 1 B coverage) -- 16 B per
                  output pixel, plus the 4 B per pixel of clearing the owner map = 20 B;
      preprocess  mm_set_target_u8 / mm_set_library_u8: 3 B read + 12 B written per pixel (u8 BGR -> f32 Lab / BGR),
                  and mm_set_library_u8_sized at 75 % detail: the general INTER_AREA resize of the u8 library
                  (3 B read per source pixel + 3 B written per detail pixel) followed by the conversion.
    Kernel time = HIP events around the kernels on the engine's stream (mm_timing.build_ms / preprocess_ms), averaged
    over --steps calls; `ms_per_step` is the whole call as the host sees it (staging copies included)."""
    import torch  # noqa: F401  (one HIP runtime per process)
    from mosaicmagnifique_amd.engine import Engine
    w = build_workload(args.config, 1)
    engine = Engine(0)
    t, rows, cols, L, size = w["t"], w["rows"], w["cols"], w["L"], w["size"]
    lines = []

    def line(name, kernel_ms, call_ms, nbytes, note):
        gbs = nbytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        lines.append({"stage": name, "kernel_ms": kernel_ms, "call_ms": call_ms, "algorithmic_bytes": nbytes,
                      "achieved_gbs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "note": note})

    if args.stage == "preprocess":
        for name, fn, nbytes, note in (
            ("preprocess_u8 target %dx%d" % (cols, rows), lambda: engine.set_target_u8(w["img_u8"], t), rows * cols * 15,
             "3 B read + 12 B written per pixel"),
            ("preprocess_u8 library %d x %dx%d" % (L, size, size), lambda: engine.set_library_u8(w["lib_u8"], t), L * size * size * 15,
             "3 B read + 12 B written per pixel"),
            ("resize_area_u8 + preprocess_u8 library at 75 %% detail (%d -> %d px)" % (size, size * 3 // 4),
             lambda: engine.set_library_u8_sized(w["lib_u8"], t, size * 3 // 4),
             L * (size * size * 3 + (size * 3 // 4) ** 2 * (3 + 3 + 12)),
             "resize: 3 B per source pixel read + 3 B per detail pixel written; conversion: 3 B read + 12 B written per detail pixel"),
        ):
            for _ in range(args.warmup):
                fn()
            k_ms, t0 = 0.0, time.perf_counter()
            for _ in range(args.steps):
                fn()
                k_ms += engine.timing()["preprocess_ms"]
            line(name, k_ms / args.steps, (time.perf_counter() - t0) / args.steps * 1e3, nbytes, note)
    else:
        stage(engine, w)
        best, _ = engine.generate(w["cells"], w["grid_w"], w["grid_h"], t, *w["rep"])
        engine.set_library_bgr(w["lib_u8"])
        for _ in range(args.warmup):
            engine.build_photomosaic(w["cells"], w["grid_w"], w["grid_h"], best, rows, cols, want_output=False)
        k_ms, t0 = 0.0, time.perf_counter()
        for _ in range(args.steps):
            engine.build_photomosaic(w["cells"], w["grid_w"], w["grid_h"], best, rows, cols, want_output=False)
            k_ms += engine.timing()["build_ms"]
        line("buildPhotomosaic %dx%d, %d cells (owner map + compose)" % (cols, rows, len(w["cells"])), k_ms / args.steps,
             (time.perf_counter() - t0) / args.steps * 1e3, rows * cols * 20,
             "4 B owner cleared + 4 B owner raised (atomicMax) + 4 B owner read + 3 B library read + 4 B BGRA + 1 B coverage written per pixel")
    head = lines[0] if args.stage == "build" else lines[1]
    emit(json.dumps({
        "metric": "achieved HBM GB/s of the %s stage" % args.stage, "value": head["achieved_gbs"], "unit": "GB/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["call_ms"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "config %s, stage %s: %s" % (w["name"], args.stage, head["stage"])},
        "roofline": {"bound": "hbm", "achieved": head["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": head["frac_of_hbm_peak"], "traffic": None, "kernel_ms": head["kernel_ms"]},
        "stages": lines,
        "hbm_stream_read_measured_gbs": engine.stream_read_gbs(4 << 30, 5),
    }))


def sweep_bench(args, emit):
    """--sweep: the shape of the reference's own benchmark harness (Benchmark/Benchmark_Generator.h:107-183) on
    synthetic inputs: a 5000 x 4000 main image (its "edgar-perez" photograph), a 214-image library (the count of
    Library/lib-images.txt) at the cell size, square cells; one parameter varied per block from the harness's
    defaults (CIE76, detail 20 %, size steps 0, cell size 128): colour difference x3, detail {25, 50, 75, 100} (and the
    default 20), size steps {0, 1, 2}, cell size {128, 64, 32, 16, 8}.  Per line, wall-clock ms of
      first   generateBestFits() of a freshly set-up generator -- what the harness's timer brackets: pre-processing of
              image and library (on the device here), staging, the best-fit pass over every size step;
      warm    the same call again (inputs resident: the hot path alone, incl. the Python host's per-cell geometry;
              `last_step_device_ms` = mm_generate's own first-launch-to-results time for the last size step);
      grid    GridGenerator::getGridState (entropy subdivision on the device when size steps > 0);
      build   buildPhotomosaic incl. the copy of the BGRA mosaic to the host;
      cpu     the CPU oracle (port of CPUPhotomosaicGenerator, 1 thread, early exit on) on the same pre-processed
              arrays -- the whole grid when that is below ~8 s, else a raster-order prefix scaled up (marked ~)."""
    import torch  # noqa: F401
    import oracle_ffi
    import cases
    from mosaicmagnifique_amd import GridUtility as G
    from mosaicmagnifique_amd import HIPPhotomosaicGenerator, synth
    from mosaicmagnifique_amd.cellshape import CellGroup
    orc = oracle_ffi.load()
    rows, cols, n_lib = 4000, 5000, 214
    img = synth.target_u8(rows, cols, lattice=64)
    ns_per_pair = {0: 6.0, 1: 6.0, 2: 200.0}
    libs = {}
    lines = []

    def one(block, t, detail, steps, cell):
        if cell not in libs:
            libs[cell] = synth.library_u8(n_lib, cell)
        lib = libs[cell]
        group = CellGroup(cases.square_shape(cell))
        group.setDetail(detail)
        group.setSizeSteps(steps)
        gen = HIPPhotomosaicGenerator(0)
        gen.setMainImage(img); gen.setLibrary(lib); gen.setColourDifference(t); gen.setCellGroup(group)
        t0 = time.perf_counter()
        state = gen.gridStateOnDevice() if steps > 0 else G.getGridState(group, rows, cols)
        grid_ms = (time.perf_counter() - t0) * 1e3
        gen.setGridState(state)
        t0 = time.perf_counter()
        ok = gen.generateBestFits()
        first_ms = (time.perf_counter() - t0) * 1e3
        if not ok:
            raise SystemExit("sweep: generateBestFits failed: " + gen.lastError)
        fits, scores = gen.getBestFits(), gen.getBestScores()
        gen.setGridState(state)
        t0 = time.perf_counter()
        gen.generateBestFits()
        warm_ms = (time.perf_counter() - t0) * 1e3
        device_ms = gen.timing()["total_ms"]             # mm_generate of the LAST size step, first launch to results on the host
        t0 = time.perf_counter()
        mosaic = gen.buildPhotomosaic((0, 0, 0, 0))
        build_ms = (time.perf_counter() - t0) * 1e3
        cells = [int((s >= 0).sum()) for s in state]
        # the CPU oracle on the same arrays, step by step
        gen2_main = gen.preprocessMainImage()
        lib_f = gen.preprocessLibraryImages()
        cpu_ms, estimated, same = 0.0, False, True
        for step in range(len(state)):
            if step > 0:
                lib_f = orc.shrink_library_f32(lib_f, 2)
            d = group.getCell(step, True)
            dsize = d.getSize()
            pairs_per_cell = n_lib * dsize * dsize
            budget_cells = int(8e9 / (len(state) * max(pairs_per_cell * ns_per_pair[int(t)], 1.0)))
            n_valid = cells[step]
            max_cells = -1 if budget_cells >= n_valid else max(budget_cells, 16)
            t0 = time.perf_counter()
            want, want_score, done = orc.generate_step(oracle_ffi.shape_of(group.getCell(step)), dsize, group.getDetail(), gen2_main,
                                                       lib_f, d.masks4(), int(t), 0, 0, state=state[step], max_cells=max_cells)
            dt = (time.perf_counter() - t0) * 1e3
            if done < n_valid:
                estimated = True
                dt *= n_valid / max(done, 1)
            cpu_ms += dt
            checked = G.cell_records(group, state[step], rows, cols, step)[:done]
            same = same and bool((fits[step][checked["grid_y"], checked["grid_x"]] == want[checked["grid_y"], checked["grid_x"]]).all())
        line = {"block": block, "diff_type": TYPE_NAMES[int(t)], "detail_percent": detail, "size_steps": steps, "cell_size": cell,
                "detail_cell": group.getCellSize(0, True), "cells_per_step": cells, "grid_ms": grid_ms, "first_ms": first_ms,
                "warm_ms": warm_ms, "last_step_device_ms": device_ms, "build_ms": build_ms, "cpu_ms": cpu_ms, "cpu_estimated_from_prefix": estimated,
                "speedup_first": cpu_ms / first_ms, "speedup_warm": cpu_ms / warm_ms, "indices_equal_on_cpu_cells": same,
                "mosaic_alpha_covered": float((mosaic[..., 3] == 255).mean())}
        lines.append(line)
        sys.stderr.write("%-12s %-13s detail %3d%% steps %d cell %3d (%2d px) cells %-22s grid %8.1f first %8.1f warm %8.2f (device %6.2f) build %7.1f cpu %s%10.0f ms  x%-8.0f %s\n" % (
            block, TYPE_NAMES[int(t)], detail, steps, cell, line["detail_cell"], cells, grid_ms, first_ms, warm_ms, device_ms, build_ms,
            "~" if estimated else " ", cpu_ms, line["speedup_first"], "ok" if same else "MISMATCH"))
        del gen

    one("(warm-up)", 1, 20, 0, 128)            # device initialisation and code loading stay out of the first line
    del lines[:]
    for t in (0, 1, 2):
        one("colour diff", t, 20, 0, 128)
    for detail in (25, 50, 75, 100):
        one("detail", 1, detail, 0, 128)
    # (detail 25 % here: at the harness's 20 % the detail cell is 25 px and the reference's own per-step sizes drift apart --
    # batchResizeMat halves the library to round(12.5) = 13 px against a detail cell of int(64 * 0.2) = 12 px)
    for steps in (0, 1, 2):
        one("size steps", 1, 25, steps, 128)
    for cell in (128, 64, 32, 16, 8):
        one("cell size", 1, 20, 0, cell)
    base = lines[1]
    emit(json.dumps({
        "metric": "ms per generateBestFits, reference harness defaults (5000x4000, 214 images, cell 128, detail 20 %, CIE76), cold call",
        "value": base["first_ms"], "unit": "ms", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": base["first_ms"],
        "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32 filter + f64 arbiter", "data": "synthetic",
        "config": {"workload": "sweep after Benchmark/Benchmark_Generator.h:107-183: 5000x4000 target, 214-image library, square cells; "
                               "colour difference x3 | detail 25..100 | size steps 0..2 | cell size 128..8"},
        "sweep": lines,
    }))


def launch_ranks(n, emit, argv=None):
    """`python3 bench.py --gpus N` without a launcher around it: run this script under `python -m torch.distributed.run`
    with N ranks (one per GPU, rendezvous on 127.0.0.1 at a free port) as a child process, pass its single JSON line
    on.  Nothing in this process has initialised the GPU when it gets here (device_count() does not)."""
    import socket
    import subprocess
    rehearsal = os.environ.get("MOSAIC_BENCH_REHEARSAL") == "1"
    if not rehearsal:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("--gpus %d: this node shows %d GPU(s) (MOSAIC_BENCH_REHEARSAL=1 runs the N-rank path on one "
                             "GPU over gloo, timings void)" % (n, have))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += list(sys.argv[1:] if argv is None else argv)
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)       # stderr passes through
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    if lines:
        emit(lines[-1])
    if p.returncode != 0 or not lines:
        raise SystemExit(p.returncode or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="D", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1: weak = one config band per rank (default for D), strong = the fixed cell list "
                         "cut into N contiguous shards (default for E)")
    ap.add_argument("--bands", type=int, default=0,
                    help="weak scaling: bands of the config stacked into one mosaic (default: one per rank); with one rank it "
                         "gives the single-GPU result an N-rank weak run must reproduce")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity legs (and the floors / first-call legs: "
                                                         "nothing but the timed steps reaches the device, as the PMC passes want it)")
    ap.add_argument("--no-floors", action="store_true", help="skip the floor legs (early exit off; fp64 only) and the first-call leg")
    ap.add_argument("--configs", default=None, metavar="B,C,...",
                    help="further configurations run after the headline in the same process, each against its whole-grid CPU golden "
                         "(default with the default config D on one GPU: B,C,E,E8,G,PD -- every BASELINE configuration, the GUI's default "
                         "cell size, and photographs at config D's scale; 'none' skips them)")
    ap.add_argument("--mcs", default=None, help="cell shape from a MosaicMagnifique .mcs file (resized to the config's cell size)")
    ap.add_argument("--mil", default=None, help="library from a MosaicMagnifique .mil file (resized to the config's cell size)")
    ap.add_argument("--stage", choices=("build", "preprocess"), default=None,
                    help="time the byte kernels either side of the hot path instead of it (buildPhotomosaic / the u8 -> f32 "
                         "pre-processing) and report their GB/s against HBM peak")
    ap.add_argument("--sweep", action="store_true",
                    help="the reference's benchmark harness shape (Benchmark/Benchmark_Generator.h): end-to-end ms per "
                         "generateBestFits beside the CPU oracle's over colour difference / detail / size steps / cell size")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="engine tuning option (mm_set_option), for experiments")
    args = ap.parse_args()

    # Only the JSON line may reach stdout: RCCL prints a version banner there when it initialises,
    # so everything else written to fd 1 during the run is sent to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(line, flush=True)

    if args.stage:
        return stage_bench(args, emit)
    if args.sweep:
        return sweep_bench(args, emit)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python3 bench.py --gpus N` (how the driver runs N = 1): start the N ranks ourselves as FRESH child
        # processes of torch.distributed.run -- before this process has touched a GPU, and never by exec -- relay
        # rank 0's JSON line and leave with the launcher's exit code
        return launch_ranks(args.gpus, emit)
    if world != args.gpus:
        args.gpus = world
    scaling = args.scaling or ("strong" if args.config in ("E", "Es") else "weak")

    import torch
    from mosaicmagnifique_amd.dist import ShardedBestFit
    from mosaicmagnifique_amd.engine import Engine

    # MOSAIC_BENCH_REHEARSAL=1: every rank drives GPU 0 and the collective runs over gloo -- the N > 1
    # code path of this script on a one-GPU box (RCCL refuses two ranks on one device).  Timings of such
    # a run mean nothing (the ranks share the device); the JSON line says so.
    rehearsal = os.environ.get("MOSAIC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # MOSAIC_BENCH_FORCE_DIST=1 runs the multi-GPU code path (RCCL init, device-side
    # candidate lists, all-gather, resolve from device memory) even with one rank.
    use_dist = world > 1 or os.environ.get("MOSAIC_BENCH_FORCE_DIST") == "1"
    comm_init_s = 0.0
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        t_comm = time.perf_counter()
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()                          # builds the communicator before anything is timed
        comm_init_s = time.perf_counter() - t_comm

    bands = args.bands or world
    w = build_workload(args.config, world, scaling, args.mcs, args.mil, bands)
    engine = Engine(local_rank)
    for item in args.opt:
        name, value = item.split("=")
        engine.set_option(name, int(value))
    t_stage = time.perf_counter()
    stage(engine, w)
    stage_s = time.perf_counter() - t_stage

    cells, gw, gh, t, rep = w["cells"], w["grid_w"], w["grid_h"], w["t"], w["rep"]
    sharded = ShardedBestFit(engine, rank, world) if use_dist else None

    from mosaicmagnifique_amd.grid import CELL_DTYPE
    cells = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
    out_arrays = (np.empty((gh, gw), dtype=np.int64), np.empty((gh, gw), dtype=np.float64))

    def step():
        if sharded is not None:
            return sharded.generate(cells, gw, gh, t, *rep)
        return engine.generate(cells, gw, gh, t, *rep, out=out_arrays)    # caller-owned result arrays, as a C++ host has

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        best, score = step()
    fence()
    if sharded is not None:
        sharded.reset_phases()
    t0 = time.perf_counter()
    score_ms, launches, exact, visited, groups_all, dropped = 0.0, 0, 0, 0, 0, 0
    raw = []
    for _ in range(args.steps):
        best, score = step()
        raw.append(engine.timing_raw())          # one small ctypes call; summed up after the timed region
    fence()
    elapsed = time.perf_counter() - t0
    for r in raw:
        score_ms += r.score_ms
        launches += r.score_launches
        exact += r.exact_rescored
        visited += r.visited_groups
        groups_all += r.total_groups
        dropped += r.bound_dropped
    tm = raw[-1].as_dict() if raw else engine.timing()
    if score_ms == 0.0 and sharded is None:
        # a small job runs without events between its phases (option "phase_timing" -1: an event costs it more than a
        # kernel): the dominant kernel's launch duration comes from three extra, untimed steps with the events on
        engine.set_option("phase_timing", 1)
        launches = 0
        for _ in range(3):
            step()
            r = engine.timing_raw()
            score_ms += r.score_ms
            launches += r.score_launches
        tm_phase = engine.timing()
        engine.set_option("phase_timing", -1)
        tm.update({k: tm_phase[k] for k in ("stage_ms", "seed_ms", "score_ms", "audit_ms", "rescore_ms", "select_ms", "resolve_ms")})
        launches_scale = args.steps / 3.0
    else:
        launches_scale = 1.0
    dist_phase_ms = None
    if sharded is not None:
        # per-step wall time of this rank's phases; over the ranks: the maximum (what the step waits for) and, for the
        # scoring, the minimum beside it (their gap is the load imbalance the all-gather then absorbs as waiting)
        ph = sharded.phase_ms
        mine = [ph[k] / max(args.steps, 1) for k in ("score", "allgather", "resolve", "deepen")]
        hi = torch.tensor(mine, dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        lo = hi.clone()
        if world > 1:
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist_phase_ms = dict(zip(("score", "allgather", "resolve", "deepen"), [round(float(v), 4) for v in hi.tolist()]))
        dist_phase_ms["score_min"] = round(float(lo[0]), 4)
        dist_phase_ms["allgather_min"] = round(float(lo[1]), 4)
        dist_phase_ms["resolve_rounds_per_step"] = ph["rounds"] / max(args.steps, 1)
        dist_phase_ms["note"] = ("host wall ms per step, maximum over the ranks (score_min / allgather_min: the minimum); allgather "
                                 "includes waiting for the slowest rank's lists; resolve = the identical verifying resolve on every rank")
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    per_rank_score_ms, ranks_seen, backend = None, None, None
    if use_dist:
        # what the communicator itself reports (a launcher that started fewer ranks than --gpus says shows here), and every
        # rank's own scoring time per step (the all-gather absorbs their spread as waiting)
        ranks_seen, backend = dist.get_world_size(), dist.get_backend()
        mine_ms = torch.tensor([sharded.phase_ms["score"] / max(args.steps, 1)], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        every = [torch.zeros_like(mine_ms) for _ in range(world)]
        if world > 1:
            dist.all_gather(every, mine_ms)
        else:
            every = [mine_ms]
        per_rank_score_ms = [round(float(v.item()), 4) for v in every]
    if rank != 0:
        dist.destroy_process_group()
        return

    n_cells = len(cells)
    engine_uses_filter = tm.get("rescore_ms", 0) > 0 or tm.get("seed_ms", 0) > 0
    value = n_cells * args.steps / elapsed
    # what ONE rank's score launches cover
    if world > 1 and scaling == "strong":
        from mosaicmagnifique_amd.dist import shard_range
        lo, hi = shard_range(n_cells, world, 0)
        my_cells = cells[lo:hi]
    else:
        my_cells = w["cells_single"] if world > 1 else cells
    alg_bytes, total_pa = algorithmic_bytes(w, my_cells)
    launches = max(launches, 1)
    per_launch_s = (score_ms / 1e3) / launches
    launches_per_step = launches * launches_scale / max(args.steps, 1)
    alg_per_launch = alg_bytes / launches_per_step
    visited_per_launch = visited / max(launches * launches_scale, 1)
    pairs_alg = total_pa * w["L"] / launches_per_step
    ciede = t == 2 and engine_uses_filter
    pretier_on = ciede and bool(tm.get("pretier"))       # (cells above 32 groups: the engine measures whether the packed-fp16 pre-tier pays on this content)

    pmc, pmc_src = pmc_record(w, args.opt, tm.get("filter_tile")) if world == 1 else (None, "PMC records are single-GPU")
    roof = {
        # CIEDE2000: the persistent two-stage kernel; RGB-Euclidean / CIE76: the LDS-resident cell tile form
        "kernel": ("filter_kernel" if t == 2 else "filter_lds_kernel") if engine_uses_filter else "score_kernel",
        "avg_launch_ms": per_launch_s * 1e3,
        "launches_per_step": launches_per_step,
        "effective_algorithmic_gbs": alg_per_launch / per_launch_s / 1e9 if per_launch_s > 0 else 0.0,
        "algorithmic_bytes_per_launch": alg_per_launch,
        "pair_evals_algorithmic_per_s": pairs_alg / per_launch_s if per_launch_s > 0 else 0.0,
        "visited_groups_per_launch": visited_per_launch,
        "visited_fraction_of_groups": visited / groups_all if groups_all else None,
        "candidates_dropped_by_block_bound": dropped / max(launches * launches_scale, 1),
        # a visited group = 256 pixel positions of one (cell, candidate): what was actually evaluated
        "pair_evals_visited_per_s": visited_per_launch * 256 / per_launch_s if per_launch_s > 0 and visited else None,
        "pmc_source": pmc_src,
        "kernel_src_sha16": kernel_source_hash(),
    }
    traffic = pmc["traffic_bytes_per_launch"] if pmc else None
    hbm = None
    if traffic is not None and per_launch_s > 0:
        hbm = {"achieved": traffic / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": traffic / per_launch_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
               "note": "fabric-side bytes (L2 misses; Infinity-Cache hits are counted) of one launch / its duration"}
    roof["traffic"] = traffic
    roof["hbm"] = hbm
    if ciede:
        # VALU-issue-bound: price the launch's VALU wave-instructions against the fp32 vector peak
        roof["bound"] = "valu"
        roof["unit"] = "TFLOP/s"
        roof["peak"] = VALU_PEAK_TFLOPS
        insts = None
        if pmc and pmc.get("SQ_INSTS_VALU_per_launch"):
            # the instruction count scales with the work actually done: tie it to this run by the
            # always-on visited-groups counter (identical when the PMC run did the same work)
            ref_visited = pmc.get("visited_groups_per_launch") or visited_per_launch
            scale = visited_per_launch / ref_visited if ref_visited else 1.0
            insts = pmc["SQ_INSTS_VALU_per_launch"] * scale
            roof["valu_wave_insts_per_launch"] = insts
            roof["visited_groups_vs_pmc_run"] = scale
            # instructions per unit of useful work (a visited group = 256 pixel pairs): what kernel work lowers;
            # `frac` below is issue-rate use, which FALLS when instructions are removed at the same cycles per instruction
            roof["valu_wave_insts_per_visited_group"] = insts / visited_per_launch if visited_per_launch else None
        if insts and per_launch_s > 0:
            rate = insts / per_launch_s
            roof["achieved"] = rate * 128 / 1e12
            roof["frac"] = rate / VALU_PEAK_WAVE_INSTS
            roof["note"] = ("VALU wave-instructions per second x 128 (fma-equivalent flop per wave64 instruction) against "
                            "the fp32 vector peak = 1024 SIMDs x 2.4 GHz / 2 cycles; transcendentals (10 of ~104 per "
                            "pixel pair) issue at quarter rate, so the mix's own ceiling is below 1 (~0.57); frac is issue-rate use, not "
                            "progress: removing instructions at the same cycles per instruction lowers it while cells/s rises")
        else:
            roof["achieved"] = None
            roof["frac"] = None
            roof["note"] = "no PMC record for this build/config: VALU rate not priced (run tools/profile.sh)"
        # WORK-NORMALISED fractions -- they rise when the kernel spends less on the same evaluated work, unlike `frac` (issue-rate
        # use, which falls when instructions are removed): (1) the pixel pairs actually evaluated x the filter form's own VALU
        # instructions (86 per pair-lane: 344 per four pairs in the ISA of the pixel loop, colour_diff_fast.hpp) against the issue
        # peak -- everything else the kernel issues (the bound's stages, preludes, exit tests, addresses) is overhead this fraction
        # does not credit; (2) the bytes of the library and cell groups it visited against the read-stream ceiling measured on
        # this box (filled in below).  SURVEY 8(d)'s own fraction -- algorithmic bytes / kernel time / 8 TB/s -- is stated beside
        # them: above 1 because the kernel provably does not do the work that byte model counts (the reference's own early exit,
        # CPUPhotomosaicGenerator.cpp:149-161, in its device form).
        if visited_per_launch and per_launch_s > 0:
            pairs_s = visited_per_launch * 256 / per_launch_s
            roof["work_normalised"] = {
                "executed_pair_evals_per_s": pairs_s,
                "form_valu_insts_per_pair": FORM_VALU_PER_PAIR,
                "useful_issue_frac": pairs_s / 64.0 * FORM_VALU_PER_PAIR / VALU_PEAK_WAVE_INSTS,
                "visited_bytes_per_launch": visited_per_launch * 4096.0,
                "visited_gbs": visited_per_launch * 4096.0 / per_launch_s / 1e9,
                "note": "useful_issue_frac = executed pair-evals / 64 lanes x 86 VALU instructions of the filter form / (1024 SIMDs x 2.4 GHz / 2); "
                        "visited bytes = visited groups x 4 KB of the library image's interleaved planes (the cell's 4 KB per visit come from the XCD's L2: "
                        "the PMC record's fabric traffic equals the library bytes alone); "
                        "visited_stream_frac = visited_gbs / the read-stream rate measured in this run",
            }
        roof["section_8d_frac"] = {
            "value": alg_per_launch / per_launch_s / 1e9 / HBM_PEAK_GBS if per_launch_s > 0 else None,
            "why_above_1": "algorithmic bytes (every pixel of every candidate once per cell) / kernel time / 8 TB/s; the kernel drops %.0f %% of the "
                           "candidates on a bound before any pixel is read and visits %.1f %% of the pixel groups -- the device form of the reference's "
                           "own early exit -- so this is an effective rate, not a roofline fraction; floor_unpruned (every pair evaluated) is the "
                           "comparable figure" % (100.0 * dropped / max(launches * launches_scale, 1) / max(len(my_cells) * w["L"], 1),
                                                  100.0 * visited / groups_all if groups_all else 0.0)}
    else:
        # RGB-Euclid / fp64-only: memory-side view, from measured traffic only
        roof["bound"] = "hbm"
        roof["unit"] = "GB/s"
        roof["peak"] = HBM_PEAK_GBS
        roof["achieved"] = hbm["achieved"] if hbm else None
        roof["frac"] = hbm["frac"] if hbm else None
        roof["note"] = ("achieved = measured fabric-side traffic / launch time; without a PMC record of this build the "
                        "fraction is not reported (the algorithmic figure is an effective rate, not a fraction)")

    out = {
        "metric": "cells matched/sec (whole node)",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling if world > 1 else "weak",
        "vs_baseline": None,
        # (every reported score and every index decision is the f64 tier's; the f32 filter only selects its contenders, and on
        # large CIEDE2000 jobs a packed-f16 pre-tier takes the filter's early-exit decisions: engine option "pretier")
        "dtype": ("f16 pre-tier + f32 filter + f64 arbiter" if pretier_on else "f32 filter + f64 arbiter") if engine_uses_filter else "f64",
        "data": "synthetic (REHEARSAL: all ranks on one GPU over gloo, timings void)" if rehearsal else
                "photographs (derived fixture of the reference's SampleImages, tests/golden/photo_*)" if w["name"] in ("P", "PD") else
                "synthetic" if not (args.mcs or args.mil) else "synthetic target; %s" % ", ".join(
            os.path.basename(p) for p in (args.mcs, args.mil) if p),
        "config": {
            "workload": "config %s: %dx%d target%s, %dx%d %s cells, %d-image library, %s, repeat range %d (+%d)" % (
                w["name"], w["cols"], w["rows"] * (bands if scaling == "weak" else 1),
                " (%d bands of %d rows)" % (bands, w["rows"]) if bands > 1 and scaling == "weak" else "",
                w["size"], w["size"], w["shape_name"], w["L"], TYPE_NAMES[t], rep[0], rep[1]),
            "cells": n_cells,
            "library_images": w["L"],
            "parallelism": ("cell-shard x%d (%s) + 1 all-gather" % (world, scaling)) if world > 1 else "single GPU",
        },
        "roofline": roof,
        "stage_once_s": stage_s,
        "comm_init_s": comm_init_s,
        "exact_rescored_per_step": exact / max(args.steps, 1),
        # sha256 over the best-fit grid (int64) and the scores (f64, NaN where the grid has no value) of the last step:
        # what an N-rank run must reproduce bit for bit (tests/test_gpu_bench_ranks.py)
        "result_digest": result_digest(best, score),
        "last_step_device_ms": {k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items()},
    }
    if dist_phase_ms is not None:
        dist_phase_ms["score_per_rank"] = per_rank_score_ms
        out["dist_phase_ms"] = dist_phase_ms
        out["rccl_ranks_seen"] = ranks_seen
        out["collective_backend"] = "%s (%s)" % (backend, "RCCL over xGMI" if backend == "nccl" else "rehearsal: gloo")
    if world == 1 and sharded is None and not (args.no_cpu or args.no_floors):
        out.update(floor_legs(engine, step, n_cells, value))
        out["first_call"] = first_call_leg(w, local_rank, args.opt)

    if world == 1:
        # what a plain coalesced read stream reaches on this device (4 GiB, past the Infinity Cache)
        out["roofline"]["hbm_stream_read_measured_gbs"] = engine.stream_read_gbs(4 << 30, 5)
        wn = out["roofline"].get("work_normalised")
        if wn:
            wn["visited_stream_frac"] = wn["visited_gbs"] / out["roofline"]["hbm_stream_read_measured_gbs"]
    if world > 1 and not args.no_cpu:
        # An N-rank line checks ITSELF (the reference's standard is cell-by-cell equality, test/tst_CUDAGenerator.h:163-212): the
        # gathered grid against the committed CPU golden -- strong shards: the whole grid; weak bands: band 0 (every band without a
        # repeat penalty) -- and, with a repeat penalty, two full grid rows INSIDE the last band evaluated by the CPU oracle as the
        # raster loop reaches them given the engine's earlier picks (the rows above them included: the window crosses no seam there,
        # and a band's rows equal the oracle's given the rows above by induction over the raster order).
        img, lib = host_arrays(engine, w)
        gold = golden_parity(w, img, lib, best, score, bands)
        if gold is not None:
            out["parity_on_golden"] = gold
        if scaling == "weak" and bands > 1 and rep[0] > 0:
            gh1 = w["state_single"].shape[0]
            band_rows = gh1 - 2
            b = bands - 1
            rows = np.unique(w["cells_single"]["grid_y"])
            first = int(np.random.default_rng(0xC0FFEE + w["L"] + b).integers(min(rep[0] + 1, len(rows) - 2), max(len(rows) - 2, rep[0] + 2)))
            first = min(first, len(rows) - 2)
            par = w["cells_single"][(w["cells_single"]["grid_y"] == rows[first]) | (w["cells_single"]["grid_y"] == rows[first + 1])][:256]
            band_view = dict(w, state_single=w["state_single"])
            sub_best = best[b * band_rows:b * band_rows + gh1]
            pos, picks, pscores, cores, dt_all = cpu_parity_rows(band_view, img, lib, sub_best, par)
            got = sub_best[pos[:, 0], pos[:, 1]]
            prel = np.abs(pscores - score[b * band_rows:b * band_rows + gh1][pos[:, 0], pos[:, 1]]) / np.maximum(pscores, 1e-300)
            out["parity_on_cpu_sample"] = {
                "indices_equal": bool(np.array_equal(picks, got)), "max_rel_score_err": float(prel.max()), "cells": int(len(par)),
                "sample": "two full grid rows (%d, %d) inside band %d of %d, each cell evaluated by the oracle as the reference's raster "
                          "loop reaches it given the engine's picks above and left of it (%d host threads, %.1f s)" % (
                              int(rows[first]), int(rows[first + 1]), b, bands, cores, dt_all)}
    if world == 1 and not args.no_cpu:
        img, lib = host_arrays(engine, w)
        grid, cscore, done, dt = cpu_baseline(w, img, lib)
        recs = my_cells[:done]
        same = all(best[r["grid_y"], r["grid_x"]] == grid[r["grid_y"], r["grid_x"]] for r in recs)
        rel = max(abs(score[r["grid_y"], r["grid_x"]] - cscore[r["grid_y"], r["grid_x"]])
                  / cscore[r["grid_y"], r["grid_x"]] for r in recs)
        out["cpu_baseline"] = {
            "value": done / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "first %d valid cells in raster order of the same workload (oracle/oracle.c, "
                      "sequential, early exit on, %.1f s)" % (done, dt),
        }
        out["speedup_vs_cpu"] = value / (done / dt)
        # the whole grid against the committed CPU golden, where there is one for this workload
        gold = golden_parity(w, img, lib, best, score, bands)
        if gold is not None:
            out["parity_on_golden"] = gold
            w["golden_ok"] = bool(gold.get("indices_equal")) and bands == 1
        # parity where the repeat window is exercised + the all-cores rate: two full grid rows
        par_recs, par_text = parity_sample(w, my_cells)
        n_par = len(par_recs)
        pos, picks, pscores, cores, dt_all = cpu_parity_rows(w, img, lib, best, par_recs)
        got = best[pos[:, 0], pos[:, 1]]
        prel = np.abs(pscores - score[pos[:, 0], pos[:, 1]]) / np.maximum(pscores, 1e-300)
        out["parity_on_cpu_sample"] = {
            "indices_equal": bool(same and np.array_equal(picks, got)),
            "max_rel_score_err": float(max(rel, prel.max())),
            "cells": int(n_par),
            "sample": (par_text % n_par) + ", each evaluated by the oracle as the reference's raster loop reaches it given "
                      "the earlier picks; plus the %d cells of the sequential 1-thread leg" % done,
        }
        out["cpu_baseline_all_cores"] = {
            "value": n_par / dt_all, "unit": "cells/s", "cores": cores, "kind": "port",
            "sample": "the same %d cells, one thread per host core of this process, repeat penalty from the finished grid "
                      "(oracle/oracle.c, early exit on, %.1f s)" % (n_par, dt_all),
        }
        out["speedup_vs_cpu_all_cores"] = value / (n_par / dt_all)
    if world == 1 and sharded is None and not args.no_cpu:
        names = args.configs if args.configs is not None else ("B,C,E,E8,G,PD" if args.config == "D" and bands == 1 and not (args.mcs or args.mil) else "none")
        names = [n for n in names.split(",") if n and n != "none"]
        if names:
            engine.close()                       # (its workspaces are parked for the legs' engines)
            out["configs"] = config_legs(local_rank, args.opt, names)
    emit(json.dumps(out))
    if use_dist:
        os.dup2(2, 1)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
