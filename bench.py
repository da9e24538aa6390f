#!/usr/bin/env python3
"""Headline benchmark: cells matched / second on the photomosaic best-fit path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config D|B|C|A|E8]

A "step" is one full pass of the hot path (generateBestFits for one size step)
over the whole cell batch, with library, masks and target already resident in
HBM.  Default workload = BASELINE.json config D, the one the north-star target is
quoted on: 4096x4096 target, 64x64 square cells (4096 cells), 5000-image library,
CIEDE2000, repeat range 4 (+500).  N > 1 (launched by torch.distributed.run, one
rank per GPU over RCCL) is weak scaling: every rank scores one config-D band of
cells (the target is the config-D image repeated per band, so the job is a
4096 x 4096*N mosaic), one all-gather of the candidate lists, identical resolve
on every rank.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel
(score_kernel): algorithmic bytes per launch (sum over cells of
L*Pa*12 + Pa*12 + Pa) over its average launch time from HIP events on the
engine's stream.  `cpu_baseline` is the CPU oracle (a port of the reference's
single-threaded generator, early exit on) timed on a bounded prefix of the same
workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable

CONFIGS = {
    # name: (rows, cols, cell, shape, L, diff type, repeat range, addition, cpu sample cells)
    "A": (512, 512, 32, "square", 100, 0, 0, 0, 256),
    "B": (2048, 2048, 64, "square", 1000, 0, 0, 0, 256),
    "C": (2048, 2048, 64, "hexagon", 1000, 2, 0, 0, 48),
    "D": (4096, 4096, 64, "square", 5000, 2, 4, 500, 8),
    "E8": (1056, 8192, 48, "square", 10000, 2, 0, 0, 4),     # one rank's band of config E
}
TYPE_NAMES = {0: "RGB-Euclidean", 1: "CIE76", 2: "CIEDE2000"}


def build_workload(name, world):
    import cases
    from mosaicmagnifique_amd import GridUtility as G
    from mosaicmagnifique_amd import synth
    rows, cols, size, shape_name, L, t, rng, add, cpu_cells = CONFIGS[name]
    shape = cases.hexagon_shape() if shape_name == "hexagon" else cases.square_shape(size)
    group = cases.group_of(shape)
    img = synth.preprocess(synth.target_u8(rows, cols, lattice=64), t)
    lib = synth.preprocess(synth.library_u8(L, size), t)
    state1 = G.getGridState(group, rows, cols)[0]
    cells1 = G.cell_records(group, state1, rows, cols)
    if world == 1:
        cells, grid_w, grid_h = cells1, state1.shape[1], state1.shape[0]
    else:
        # weak scaling: `world` bands, each the single-GPU image; band b's cells read the
        # same target pixels (y unchanged) but sit `b * band_rows` grid rows lower.
        if shape_name != "square":
            raise SystemExit("multi-GPU bands are defined for square cells")
        band_rows = state1.shape[0] - G.PAD_GRID
        parts = []
        for b in range(world):
            c = cells1.copy()
            c["grid_y"] += b * band_rows
            parts.append(c)
        cells = np.concatenate(parts)
        grid_w, grid_h = state1.shape[1], band_rows * world + G.PAD_GRID
    return dict(name=name, shape=shape, img=img, lib=lib, cells=cells, cells_single=cells1,
                grid_w=grid_w, grid_h=grid_h, t=t, rep=(rng, add), L=L, size=size,
                rows=rows, cols=cols, cpu_cells=cpu_cells, state_single=state1)


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


def cpu_baseline(w):
    """The oracle (port of CPUPhotomosaicGenerator, 1 thread, early exit) on the
    first `cpu_cells` valid cells in raster order of the same inputs."""
    import oracle_ffi
    o = oracle_ffi.load()
    n = w["cpu_cells"]
    t0 = time.perf_counter()
    grid, scores, done = o.generate_best_fits(
        oracle_ffi.shape_of(w["shape"]), w["img"], w["lib"], w["shape"].masks4(), w["t"],
        w["rep"][0], w["rep"][1], max_cells=n, state=w["state_single"])
    dt = time.perf_counter() - t0
    return grid, scores, done, dt


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


def cpu_baseline_all_cores(w, per_thread=2):
    """the same port on every host core of this process's CPU share -- one
    thread per core, each scoring its own `per_thread` valid cells (no repeat penalty: base
    scores are what parallelises; the penalty resolve is negligible on the CPU too).
    ctypes releases the GIL inside the oracle call."""
    import threading

    import oracle_ffi
    o = oracle_ffi.load()
    cores = host_cores()
    state = w["state_single"]
    valid = np.argwhere(state >= 0)
    cores = max(1, min(cores, len(valid) // per_thread))
    shape = oracle_ffi.shape_of(w["shape"])
    masks = w["shape"].masks4()
    done = [0] * cores

    def work(k):
        mine = np.full(state.shape, -1, dtype=np.int64)
        for gy, gx in valid[k * per_thread:(k + 1) * per_thread]:
            mine[gy, gx] = state[gy, gx]
        done[k] = o.generate_best_fits(shape, w["img"], w["lib"], masks, w["t"], 0, 0, state=mine)[2]

    threads = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    return sum(done), cores, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="D", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
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

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    from mosaicmagnifique_amd.dist import ShardedBestFit
    from mosaicmagnifique_amd.engine import Engine

    torch.cuda.set_device(local_rank)
    # MOSAIC_BENCH_FORCE_DIST=1 runs the multi-GPU code path (RCCL init, device-side
    # candidate lists, all-gather, resolve from device memory) even with one rank.
    use_dist = world > 1 or os.environ.get("MOSAIC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    w = build_workload(args.config, world)
    engine = Engine(local_rank)
    for item in args.opt:
        name, value = item.split("=")
        engine.set_option(name, int(value))
    t_stage = time.perf_counter()
    engine.set_library(w["lib"])
    engine.set_masks(w["shape"].masks4())
    engine.set_target(w["img"])
    stage_s = time.perf_counter() - t_stage

    cells, gw, gh, t, rep = w["cells"], w["grid_w"], w["grid_h"], w["t"], w["rep"]
    sharded = ShardedBestFit(engine, rank, world) if use_dist else None

    def step():
        if sharded is not None:
            return sharded.generate(cells, gw, gh, t, *rep)
        return engine.generate(cells, gw, gh, t, *rep)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        best, score = step()
    fence()
    t0 = time.perf_counter()
    score_ms, launches, exact = 0.0, 0, 0
    for _ in range(args.steps):
        best, score = step()
        tm = engine.timing()
        score_ms += tm["score_ms"]
        launches += tm["score_launches"]
        exact += tm["exact_rescored"]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    if rank != 0:
        dist.destroy_process_group()
        return

    n_cells = len(cells)
    engine_uses_filter = tm.get("rescore_ms", 0) > 0 or tm.get("seed_ms", 0) > 0
    value = n_cells * args.steps / elapsed
    my_cells = w["cells_single"]                      # what ONE rank's score launches cover
    alg_bytes, total_pa = algorithmic_bytes(w, my_cells)
    per_launch_s = (score_ms / 1e3) / max(launches, 1)
    launches_per_step = max(launches, 1) / max(args.steps, 1)
    achieved = alg_bytes / launches_per_step / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
    # HBM-side traffic of the dominant kernel from rocprofv3 PMC passes of this same
    # command (profiles/summarise_pmc.py; FETCH_SIZE doubled as the gfx950 guide says)
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        entry = json.load(open(pmc_path)).get(w["name"])
        if entry:
            traffic = entry["traffic_bytes_per_launch"]
            traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
    out = {
        "metric": "cells matched/sec (whole node)",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "config %s: %dx%d target%s, %dx%d %s cells, %d-image library, %s, repeat range %d (+%d)" % (
                w["name"], w["cols"], w["rows"] * world, " (%d bands of %d rows)" % (world, w["rows"]) if world > 1 else "",
                w["size"], w["size"], CONFIGS[w["name"]][3], w["L"], TYPE_NAMES[t], rep[0], rep[1]),
            "cells": n_cells,
            "library_images": w["L"],
            "parallelism": "cell-shard x%d + 1 all-gather" % world if world > 1 else "single GPU",
        },
        "roofline": {
            "kernel": "filter_kernel" if engine_uses_filter else "score_kernel",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes / launches_per_step,
            "avg_launch_ms": per_launch_s * 1e3,
            "launches_per_step": launches_per_step,
            "pair_evals_per_s": total_pa * w["L"] / launches_per_step / per_launch_s if per_launch_s > 0 else 0.0,
            "note": "achieved = bytes a naive stream would move / launch time (SURVEY 8d); the kernel skips pruned "
                    "pixel groups and shares tiles through L2, so measured traffic is lower and frac can pass 1: "
                    "it is VALU-bound (profiles/r01_v11_configD_pmc_SQ.csv), not HBM-bound",
        },
        "stage_once_s": stage_s,
        "exact_rescored_per_step": exact / max(args.steps, 1),
        "last_step_device_ms": {k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items()},
    }

    if world == 1:
        # what a plain coalesced read stream reaches on this device (4 GiB, past the Infinity Cache)
        measured = engine.stream_read_gbs(4 << 30, 5)
        out["roofline"]["peak_measured"] = measured
        out["roofline"]["frac_of_measured"] = achieved / measured if measured > 0 else None
    if world == 1 and not args.no_cpu:
        grid, cscore, done, dt = cpu_baseline(w)
        recs = my_cells[:done]
        same = all(best[r["grid_y"], r["grid_x"]] == grid[r["grid_y"], r["grid_x"]] for r in recs)
        rel = max(abs(score[r["grid_y"], r["grid_x"]] - cscore[r["grid_y"], r["grid_x"]])
                  / cscore[r["grid_y"], r["grid_x"]] for r in recs)
        out["cpu_baseline"] = {
            "value": done / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "first %d valid cells in raster order of the same workload (oracle/oracle.c, "
                      "early exit on, %.1f s)" % (done, dt),
        }
        out["parity_on_cpu_sample"] = {"indices_equal": bool(same), "max_rel_score_err": rel}
        out["speedup_vs_cpu"] = value / (done / dt)
        n_all, cores, dt_all = cpu_baseline_all_cores(w)
        out["cpu_baseline_all_cores"] = {
            "value": n_all / dt_all, "unit": "cells/s", "cores": cores, "kind": "port",
            "sample": "%d valid cells, %d per thread, one thread per host core of this process, base scores "
                      "only (oracle/oracle.c, early exit on, %.1f s)" % (n_all, n_all // max(cores, 1), dt_all),
        }
        out["speedup_vs_cpu_all_cores"] = value / (n_all / dt_all)
    emit(json.dumps(out))
    if use_dist:
        os.dup2(2, 1)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
