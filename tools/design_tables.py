#!/usr/bin/env python3
"""Prints the tables of DESIGN.md section 5 from the bench lines collected under profiles/ (python tools/design_tables.py r04_v40)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {
    "A": "A: 512², 32² cells, 100 images, RGB", "B": "B: 2048², 64² cells, 1000 images, RGB",
    "C": "C: 2048², Hexagon 64, 1000 images, CIEDE2000", "D": "**D: 4096², 64² cells, 5000 images, CIEDE2000, repeat 4 (+500)**",
    "E8": "E's one-rank shard (3656 cells of 48², 10 000 images)", "E_1gpu": "E whole on one GPU (29 241 cells)",
    "D_dist1": "D through the one-rank RCCL path (`MOSAIC_BENCH_FORCE_DIST=1`)",
    "F": "F: the reference GUI's defaults -- 4096², 128² cells at full detail, 1000 images, RGB", "G": "G: F's shape as CIEDE2000",
    "H": "H: the reference harness's CIEDE2000 line at its detail size -- 1280 cells of 25², 214 images",
    "I": "I: 4000², 80² cells (25 pixel groups), 1000 images, CIEDE2000",
    "P": "P: config D's parameters on PHOTOGRAPHS -- 2048 × 1363 target, 704 cells of 64², 1024 tiles",
    "PD": "**PD: config D's job on PHOTOGRAPHS -- 4096² target, 4096 cells of 64², 5000 tiles, CIEDE2000, repeat 4 (+500)**",
}


def load(tag, name):
    p = os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (tag, name))
    return json.load(open(p)) if os.path.exists(p) else None


def main():
    tag = sys.argv[1]
    print("| configuration | cells/s | ms per step | dominant kernel, ms per launch | groups visited | `roofline.frac` | fabric traffic per launch, ÷ 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    for name in ("A", "B", "C", "D", "E8", "E_1gpu", "D_dist1", "F", "G", "H", "I", "P", "PD"):
        d = load(tag, name)
        if not d:
            continue
        r = d["roofline"]
        hbm = r.get("hbm") or {}
        frac = r.get("frac")
        if r.get("launches_per_step", 1) > 1.5:       # a step of several launches of very different sizes (deepening rounds): the average launch is not a kernel
            frac, hbm = None, {}
        print("| %s | %s | **%.3g** | `%s` %.3g | %s | %s | %s |" % (
            NAMES[name], ("%.2f M" % (d["value"] / 1e6)) if d["value"] >= 1e6 else ("%.1f k" % (d["value"] / 1e3)), d["ms_per_step"],
            r["kernel"], r["avg_launch_ms"], ("%.1f %%" % (100 * r["visited_fraction_of_groups"])) if r.get("visited_fraction_of_groups") else "—",
            ("%.2f (%s)" % (frac, r["bound"])) if frac else "—",
            ("%.3g GB, %.2f" % (hbm["traffic"] / 1e9, hbm["frac"])) if hbm else "—"))
    print()
    print("| configuration | headline ms | early exit off (`prune` 0) | fp64 only (`filter` 0) | headline ÷ floors | CPU oracle, 1 thread (cells/s) | all cores | first call: fresh engine / after a destroyed one (ms) |")
    print("|---|---|---|---|---|---|---|---|")
    for name in ("A", "B", "C", "D", "E8", "F", "G", "H", "I", "P", "PD"):
        d = load(tag, name)
        if not d or "floor_unpruned" not in d:
            continue
        fc = d.get("first_call", {})
        print("| %s | %.3g | %.3g | %.4g | %.1f× / %.0f× | %.3g | %.3g (%d) | %.1f / %.2f |" % (
            name, d["ms_per_step"], d["floor_unpruned"]["ms_per_step"], d["floor_fp64_only"]["ms_per_step"],
            d["floor_unpruned"]["headline_over_floor"], d["floor_fp64_only"]["headline_over_floor"],
            d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"], d["cpu_baseline_all_cores"]["cores"],
            fc["first"]["first_call_ms"], fc["again"]["first_call_ms"]))
    print()
    for name in ("B", "C", "D"):
        d = load(tag, name)
        if d and d.get("parity_on_golden"):
            g = d["parity_on_golden"]
            print("config %s: whole grid vs golden: %d cells, indices equal %s, max rel score err %.2g; sample: %s" % (
                name, g["cells"], g["indices_equal"], g["max_rel_score_err"], d["parity_on_cpu_sample"]["sample"][:160]))
    d = load(tag, "D")
    if d:
        t = d["last_step_device_ms"]
        print("config D step:", {k: t[k] for k in ("stage_ms", "seed_ms", "score_ms", "audit_ms", "rescore_ms", "select_ms", "resolve_ms", "total_ms", "resolve_sweeps", "host_syncs")})
        print("cpu:", d["cpu_baseline"], d["speedup_vs_cpu"], d["cpu_baseline_all_cores"], d["speedup_vs_cpu_all_cores"])
        print("stream:", d["roofline"].get("hbm_stream_read_measured_gbs"))


def sweep(path):
    """The harness-shape table of DESIGN section 5 from a `bench.py --sweep` line."""
    doc = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
    print("| block | colour difference | detail | size steps | cell (detail cell) | cells per step | grid ms | first ms | warm ms (device) | build ms | cpu ms | cpu ÷ first / ÷ warm |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for ln in doc["sweep"]:
        print("| %s | %s | %d %% | %d | %d (%d px) | %s | %.1f | %.1f | %.2f (%.2f) | %.1f | %s%.0f | %.0f× / %.0f× |" % (
            ln["block"], ln["diff_type"], ln["detail_percent"], ln["size_steps"], ln["cell_size"], ln["detail_cell"],
            " + ".join(str(c) for c in ln["cells_per_step"]), ln["grid_ms"], ln["first_ms"], ln["warm_ms"], ln["last_step_device_ms"],
            ln["build_ms"], "~" if ln["cpu_estimated_from_prefix"] else "", ln["cpu_ms"], ln["speedup_first"], ln["speedup_warm"]))
    print("all indices equal on the CPU's cells:", all(ln["indices_equal_on_cpu_cells"] for ln in doc["sweep"]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--sweep":
        sweep(sys.argv[2])
    else:
        main()
