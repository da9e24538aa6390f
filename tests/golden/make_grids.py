#!/usr/bin/env python3
"""Whole-grid CPU goldens at the benchmarked sizes (test infrastructure; CPU only, no GPU, no reference needed).

Runs the CPU oracle (oracle/oracle.c: the restatement of CPUPhotomosaicGenerator::generateBestFits /
findCellBestFit / calculateRepeats, CPUPhotomosaicGenerator.cpp:33-225) over the WHOLE grid of BASELINE.json's
configurations B, C, D and E on exactly the arrays the tests build (tests/cases.py `inputs`, i.e. synth.py's
counter-hash images and its numpy Lab conversion), and stores per configuration

    tests/golden/grid_<config>.npz   best  int32 [gridH][gridW]   (-1 = nullopt)
                                     score f64   [gridH][gridW]   (bestVariant of the pick; NaN where nullopt)
                                     meta  JSON: sizes, sha256 of the input arrays, how long it took, the mode

so that `-m gpu` tests compare the ENTIRE grid of the engine with the reference's algorithm, the way the
reference's own differential tests do (test/tst_CUDAGenerator.h:163-212, CompareBestFits test/tst_Generator.h:25-63).

Modes
* no repeat penalty (B, C, E): cells are independent (calculateRepeats returns an empty map at range 0), so the
  raster loop's pick for a cell is orc_picks_given_grid on an empty grid -- spread over --threads host threads.
* with the penalty (D): `--mode sequential` is ONE call of orc_generate_best_fits (the raster loop itself, one
  thread, ~70 min for config D); `--mode wavefront` evaluates the cells in the order t = x + (R+1)*y: every cell
  the window of calculateRepeats reads (rows y-R..y-1 up to column x+R, and the same row left of x) has a smaller t,
  so the cells of one t are independent given the earlier ones and each is picked by orc_picks_given_grid exactly as
  the raster loop reaches it -- the same grid by induction, on all host threads.  `--check` re-runs the other
  mode's result against a stored file.

    python tests/golden/make_grids.py B C            # seconds / minutes
    python tests/golden/make_grids.py D --mode sequential
    python tests/golden/make_grids.py E --threads 8  # ~1 h
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, HERE)

# name: rows, cols, cell size, shape, library size, difference type, repeat range, repeat addition
CONFIGS = {
    "B": (2048, 2048, 64, "square", 1000, 0, 0, 0),
    "C": (2048, 2048, 64, "hexagon", 1000, 2, 0, 0),
    "D": (4096, 4096, 64, "square", 5000, 2, 4, 500),
    "E": (8192, 8192, 48, "square", 10000, 2, 0, 0),
    # bench.py's further configurations: the reference's CPU-runnable case (A), the GUI's default cell size as RGB and as
    # CIEDE2000 (F, G: 128 px -- where the packed-fp16 pre-tier is on), its harness's CIEDE2000 line (H), 80-px cells (I),
    # photographs with config D's parameters (P: tests/golden/photo_*.png)
    "A": (512, 512, 32, "square", 100, 0, 0, 0),
    "F": (4096, 4096, 128, "square", 1000, 0, 0, 0),
    "G": (4096, 4096, 128, "square", 1000, 2, 0, 0),
    "H": (800, 1000, 25, "square", 214, 2, 0, 0),
    "I": (4000, 4000, 80, "square", 1000, 2, 0, 0),
    "P": (1363, 2048, 64, "square", 1024, 2, 4, 500),
    # ... and at config D's scale (PD: tests/golden/photo_pd_*.jpg, make_photo_fixture_pd.py): 4096 cells, 5000 tiles, repeat range 4
    "PD": (4096, 4096, 64, "square", 5000, 2, 4, 500),
    # the reduced shapes of D and E that bench.py's rank rehearsals run (tests/test_gpu_bench_ranks.py): an N-rank line checks itself
    # against these (band 0 of the weak-scaled Ds, the whole grid of the strong-scaled Es)
    "Ds": (1024, 1024, 64, "square", 400, 2, 4, 500),
    "Es": (1000, 1000, 48, "square", 500, 2, 0, 0),
    # reduced shapes (seconds): used by the CPU suite to hold this script's two modes to each other
    "d": (512, 512, 32, "square", 60, 2, 4, 500),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def scenario(name):
    import cases
    from mosaicmagnifique_amd import GridUtility as G
    rows, cols, size, shape_name, L, t, rng, add = CONFIGS[name]
    shape = cases.hexagon_shape() if shape_name == "hexagon" else cases.square_shape(size)
    if name in ("P", "PD"):
        from mosaicmagnifique_amd import synth
        if name == "P":
            import make_photo_fixture
            img_u8, lib_u8 = make_photo_fixture.load()
        else:
            import make_photo_fixture_pd
            img_u8, lib_u8 = make_photo_fixture_pd.load()
        img, lib = synth.preprocess(img_u8, t), synth.preprocess(lib_u8, t)
    else:
        img, lib = cases._inputs(rows, cols, size, L, t, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    return dict(name=name, rows=rows, cols=cols, size=size, shape=shape, L=L, t=t, rep=(rng, add),
                img=img, lib=lib, state=state, cells=cells)


def golden_path(name):
    return os.path.join(HERE, "grid_%s.npz" % name)


def run(name, mode, threads, log=print):
    import oracle_ffi
    o = oracle_ffi.load()
    s = scenario(name)
    sh, masks = oracle_ffi.shape_of(s["shape"]), s["shape"].masks4()
    cells, state = s["cells"], s["state"]
    gh, gw = state.shape
    rng, add = s["rep"]
    log("config %s: %d valid cells on a %d x %d padded grid, %d images, repeat %s" % (name, len(cells), gw, gh, s["L"], s["rep"]))
    t0 = time.perf_counter()
    if mode == "sequential":
        grid, scores, done = o.generate_best_fits(sh, s["img"], s["lib"], masks, s["t"], rng, add, state=state)
        assert done == len(cells)
        best = grid
        score = np.where(state >= 0, scores, np.nan)
    else:
        best = np.full((gh, gw), -1, dtype=np.int64)
        score = np.full((gh, gw), np.nan)
        order = cells["grid_x"].astype(np.int64) + (rng + 1) * cells["grid_y"].astype(np.int64) if rng > 0 \
            else np.arange(len(cells), dtype=np.int64) // max(1, 32 * threads)
        steps = np.unique(order)
        done = 0
        last = t0
        for t in steps:
            sel = cells[order == t]
            pos = np.stack([sel["grid_y"], sel["grid_x"]], axis=1)
            picks, sc = o.picks_given_grid(sh, s["img"], s["lib"], masks, s["t"], rng, add, best, pos, threads=threads)
            best[pos[:, 0], pos[:, 1]] = picks
            score[pos[:, 0], pos[:, 1]] = np.where(picks >= 0, sc, np.nan)
            done += len(sel)
            now = time.perf_counter()
            if now - last > 60:
                last = now
                log("  %d / %d cells, %.0f s" % (done, len(cells), now - t0))
    dt = time.perf_counter() - t0
    valid = state >= 0
    assert (best[valid] >= 0).all() and (best[~valid] == -1).all()
    meta = dict(config=name, rows=s["rows"], cols=s["cols"], cell_size=s["size"], library=s["L"], diff_type=s["t"],
                repeat=list(s["rep"]), cells=int(len(cells)), grid=[int(gh), int(gw)], mode=mode, threads=threads,
                seconds=round(dt, 1), img_sha256=sha(s["img"]), lib_sha256=sha(s["lib"]),
                oracle="oracle/oracle.c orc_generate_best_fits / orc_picks_given_grid, early exit on",
                numpy=np.__version__)
    log("  done in %.0f s (%.2f cells/s)" % (dt, len(cells) / dt))
    return best.astype(np.int32), score, meta


def load(name):
    """(best int64 [gh][gw] with -1 = nullopt, score f64, meta dict) of a stored golden, or None."""
    p = golden_path(name)
    if not os.path.exists(p):
        return None
    z = np.load(p)
    return z["best"].astype(np.int64), z["score"], json.loads(str(z["meta"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+", choices=sorted(CONFIGS))
    ap.add_argument("--mode", choices=["sequential", "wavefront"], default="wavefront")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--check", action="store_true", help="compare with the stored file instead of writing it")
    args = ap.parse_args()
    for name in args.configs:
        best, score, meta = run(name, args.mode, args.threads)
        if args.check:
            old = load(name)
            assert old is not None, "no stored golden for " + name
            same = np.array_equal(old[0], best) and np.array_equal(old[1], score, equal_nan=True)
            print("config %s: %s mode %s the stored grid (%s mode)" % (name, args.mode, "REPRODUCES" if same else "DIFFERS FROM", old[2]["mode"]))
            if same:
                old[2].setdefault("confirmed_by", []).append(dict(mode=args.mode, seconds=meta["seconds"], threads=args.threads))
                np.savez_compressed(golden_path(name), best=old[0].astype(np.int32), score=old[1], meta=json.dumps(old[2]))
            else:
                sys.exit(1)
        else:
            np.savez_compressed(golden_path(name), best=best, score=score, meta=json.dumps(meta))
            print("wrote %s (%d bytes)" % (golden_path(name), os.path.getsize(golden_path(name))))


if __name__ == "__main__":
    main()
