"""WHOLE-grid CPU == device at the benchmarked sizes (the reference's own standard: its differential generator tests
compare EVERY cell of the CPU and CUDA grids, test/tst_CUDAGenerator.h:163-212 through CompareBestFits,
test/tst_Generator.h:25-63).

tests/golden/grid_{B,C,D,E}.npz hold the best fit and the score of every cell of BASELINE.json's configurations as
the CPU oracle (oracle/oracle.c: the raster loop of CPUPhotomosaicGenerator.cpp:33-225, sequential fp64 sums, early
exit, the repeat window of :185-225) produced them in the build container (tests/golden/make_grids.py, which says
how: config D's 4096 cells in raster-dependency order with the full 4-rows-above repeat window).  The GPU tests build
the same inputs from the same counter hash, run the engine through the C ABI and compare the ENTIRE grid: indices
exact, scores within 1e-4 relative (CIEDE2000; north_star) or 1e-12 (RGB: same terms, other summation order).

Without a GPU: the stored grids are spot-checked against the oracle here (a seeded sample of cells per
configuration, each as the raster loop reaches it given the stored earlier picks), the inputs' digests are compared
where building them is cheap, and the generator script's two modes are held to each other on a reduced config D."""
import json
import os
import sys

import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from oracle_ffi import shape_of

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_grids  # noqa: E402

REL_TOL = {0: 1e-12, 2: 1e-4}


def golden(name):
    got = make_grids.load(name)
    if got is None:
        pytest.skip("tests/golden/grid_%s.npz not generated" % name)
    return got


@pytest.mark.parametrize("name", ["A", "B", "C", "D", "E", "F", "G", "H", "I", "P"])
def test_stored_grid_is_well_formed(name):
    best, score, meta = golden(name)
    rows, cols, size, shape_name, L, t, rng, add = make_grids.CONFIGS[name]
    shape = cases.hexagon_shape() if shape_name == "hexagon" else cases.square_shape(size)
    state = G.getGridState(cases.group_of(shape), rows, cols)[0]
    assert best.shape == state.shape == score.shape
    valid = state >= 0
    assert meta["cells"] == int(valid.sum()) and meta["library"] == L and meta["repeat"] == [rng, add]
    assert (best[valid] >= 0).all() and (best[valid] < L).all() and (best[~valid] == -1).all()
    assert np.isfinite(score[valid]).all() and np.isnan(score[~valid]).all()
    if name == "D":      # the penalty did act: far fewer equal neighbours than the base scores alone would give
        inner = best[G.PAD_GRID:-G.PAD_GRID or None, G.PAD_GRID:-G.PAD_GRID or None]
        assert (inner[:, 1:] == inner[:, :-1]).mean() < 0.01
    # (on photographs, config P, an addition of 500 is small beside the score differences: 15 % of the neighbours still repeat)


@pytest.mark.parametrize("name,n_sample", [("B", 48), ("C", 24), ("D", 6)])
def test_stored_grid_against_the_oracle_on_a_sample(oracle, name, n_sample):
    """A seeded sample of cells (clipped ones and, for D, interior cells with four rows above them included), each
    evaluated by the oracle as the reference's raster loop reaches it when the earlier cells hold the STORED picks."""
    best, score, meta = golden(name)
    s = make_grids.scenario(name)
    assert make_grids.sha(s["img"]) == meta["img_sha256"] and make_grids.sha(s["lib"]) == meta["lib_sha256"]
    cells = s["cells"]
    rng = np.random.default_rng(0x601D + ord(name))
    pick = rng.choice(len(cells), n_sample, replace=False)
    pos = np.stack([cells["grid_y"][pick], cells["grid_x"][pick]], axis=1)
    picks, scores = oracle.picks_given_grid(shape_of(s["shape"]), s["img"], s["lib"], s["shape"].masks4(), s["t"],
                                            s["rep"][0], s["rep"][1], best, pos, threads=min(8, len(os.sched_getaffinity(0))))
    assert np.array_equal(picks, best[pos[:, 0], pos[:, 1]])
    assert np.array_equal(scores, score[pos[:, 0], pos[:, 1]])


def test_generator_modes_agree_on_a_reduced_config_d():
    """make_grids.py's wavefront order (t = x + (R + 1) y, all host threads) gives the grid of the sequential raster loop."""
    seq = make_grids.run("d", "sequential", 1, log=lambda *_: None)
    wav = make_grids.run("d", "wavefront", 4, log=lambda *_: None)
    assert np.array_equal(seq[0], wav[0]) and np.array_equal(seq[1], wav[1], equal_nan=True)


# ----------------------------------------------------------------------------------------------------------------------
# the parity tests proper: the engine through the C ABI against the stored whole grids
# ----------------------------------------------------------------------------------------------------------------------

def whole_grid_against_golden(engine, name, options=None):
    best_want, score_want, meta = golden(name)
    s = make_grids.scenario(name)
    # the inputs are a pure function of the counter hash; a host whose numpy rounds `cbrt` / `pow` differently in the last
    # place would move single f32 Lab values (scores by ~1e-10 relative, picks only at ties of that size): reported, not fatal
    same_inputs = make_grids.sha(s["img"]) == meta["img_sha256"] and make_grids.sha(s["lib"]) == meta["lib_sha256"]
    shape, cells, state, L, t = s["shape"], s["cells"], s["state"], s["L"], s["t"]
    engine.set_library(s["lib"])
    engine.set_masks(shape.masks4())
    engine.set_target(s["img"])
    for k, v in (options or {}).items():
        engine.set_option(k, v)
    try:
        best, score = engine.generate(cells, state.shape[1], state.shape[0], t, *s["rep"])
        tm = engine.timing()
        # large jobs measure the fp16 pre-tier over their first two launches (off, then on) and keep the faster form: all three
        # launches must give this grid
        for _ in range(2 if name in ("D", "E", "G", "I") and not options else 0):
            again, again_score = engine.generate(cells, state.shape[1], state.shape[0], t, *s["rep"])
            assert np.array_equal(again, best) and np.array_equal(again_score, score, equal_nan=True)
    finally:
        for k in (options or {}):
            engine.set_option(k, {"audit": 1024, "spec_k": 4, "pretier": -1}.get(k, 1))
    valid = state >= 0
    assert (best[~valid] == L).all()
    got = np.where(best < L, best, -1)
    wrong = np.argwhere(got != best_want)
    assert len(wrong) == 0, "%d of %d cells differ (inputs identical to the golden's: %s); first: %s" % (
        len(wrong), int(valid.sum()), same_inputs, [(int(y), int(x), int(got[y, x]), int(best_want[y, x])) for y, x in wrong[:5]])
    rel = np.abs(score[valid] - score_want[valid]) / np.maximum(np.abs(score_want[valid]), 1e-300)
    zero = score_want[valid] == 0          # cells without an active pixel score exactly 0 on both sides
    assert (score[valid][zero] == 0).all()
    assert rel[~zero].max() <= REL_TOL[t], rel[~zero].max()
    assert tm["band_alarms"] == 0
    return tm, float(rel[~zero].max()), same_inputs


@pytest.mark.gpu
def test_config_b_whole_grid_equals_the_cpu_golden(engine):
    """Config B: 2048 x 2048, 1024 cells of 64 x 64, 1000 images, RGB-Euclidean -- every cell."""
    tm, rel, same = whole_grid_against_golden(engine, "B")
    print(json.dumps(dict(config="B", max_rel_score_err=rel, inputs_identical=same)))


@pytest.mark.gpu
def test_config_c_whole_grid_equals_the_cpu_golden(engine):
    """Config C: Hexagon.mcs mask, 1694 valid cells of which 161 clipped (22 slivers without an active pixel), CIEDE2000."""
    tm, rel, same = whole_grid_against_golden(engine, "C")
    print(json.dumps(dict(config="C", max_rel_score_err=rel, inputs_identical=same)))


@pytest.mark.gpu
def test_config_d_whole_grid_equals_the_cpu_golden(engine):
    """Config D, the headline: 4096 cells x 5000 images, CIEDE2000, repeat range 4 (+500) -- every cell, i.e. the full
    4-rows-above window of calculateRepeats (CPUPhotomosaicGenerator.cpp:185-225) in front of the oracle on 60 of the 64 rows."""
    tm, rel, same = whole_grid_against_golden(engine, "D")
    assert tm["seed_ms"] > 0 and tm["rescore_ms"] > 0          # the pruned two-tier path is what ran
    print(json.dumps(dict(config="D", max_rel_score_err=rel, inputs_identical=same, spec_rounds=tm["spec_rounds"])))


@pytest.mark.gpu
def test_config_d_whole_grid_equals_the_cpu_golden_on_the_step_by_step_path(engine):
    """... and through the host-decided path with forced deepening rounds (short lists of 2)."""
    tm, rel, same = whole_grid_against_golden(engine, "D", {"fast_path": 0, "spec_k": 2})
    engine.set_option("spec_k", 4)
    assert tm["deepened_cells"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["B", "C", "D", "F", "G"])
def test_every_dropped_candidate_audited_at_the_benchmarked_size(engine, name):
    """`audit` = 1: EVERY candidate that leaves the CIEDE2000 filter without a full score -- config D: 2 x 10^7 per launch, dropped
    by the bound's two stages, by the packed-fp16 walk (forced on here) or by the fp32 walk's exit test -- is walked once more by the
    fp32 form without the bound, and one that ended clearly below the limit would void the launch.  None does: no cell goes down
    the fp64-only path (`dense_cells` 0, `voided_launches` 0), and the grid is the CPU golden's.  (C: the
    hexagon's clipped cells; G: cells above 16 groups, stage 0 feeding the candidate loop directly; B and F: the RGB filter's
    own bound and audit_euclid_kernel, 64- and 128-px cells.)"""
    tm, rel, same = whole_grid_against_golden(engine, name, {"audit": 1, "pretier": 1})
    from mosaicmagnifique_amd.engine import has_pretier
    assert tm["pretier"] == (1 if make_grids.CONFIGS[name][5] == 2 and has_pretier() else 0) and tm["dense_cells"] == 0 and tm["voided_launches"] == 0, tm
    print(json.dumps(dict(config=name, audited="every dropped candidate", visited_groups=tm["visited_groups"], total_groups=tm["total_groups"])))


@pytest.mark.gpu
def test_config_e_whole_grid_equals_the_cpu_golden(engine):
    """Config E (BASELINE config 5) whole: 8192 x 8192, 171 x 171 cells of 48 x 48 with the clipped last row / column,
    10 000 images, CIEDE2000 -- all 29 241 cells."""
    tm, rel, same = whole_grid_against_golden(engine, "E")
    print(json.dumps(dict(config="E", max_rel_score_err=rel, inputs_identical=same)))


@pytest.mark.gpu
@pytest.mark.parametrize("name,members", [("D", 8), ("C", 3)])
def test_sharded_group_whole_grid_equals_the_cpu_golden(name, members):
    """The multi-GPU split at the benchmarked size against the CPU golden itself (not only against one engine): `members`
    engines (all on device 0 here, so the lists travel through host memory; on an 8-GPU node through ONE ncclAllGather) each
    score a contiguous block of the raster-ordered cells with short lists, the lists are gathered, every member runs the
    identical verifying resolve -- config D: eight blocks of 512 cells with the repeat window reaching across every seam --
    and the grid that comes back equals the oracle's over every cell."""
    from mosaicmagnifique_amd.engine import EngineGroup
    best_want, score_want, meta = golden(name)
    s = make_grids.scenario(name)
    state, L, t = s["state"], s["L"], s["t"]
    with EngineGroup([0] * members) as g:
        g.set_library(s["lib"]); g.set_masks(s["shape"].masks4()); g.set_target(s["img"])
        best, score = g.generate(s["cells"], state.shape[1], state.shape[0], t, *s["rep"])
    got = np.where(best < L, best, -1)
    assert np.array_equal(got, best_want), int((got != best_want).sum())
    valid = state >= 0
    rel = np.abs(score[valid] - score_want[valid]) / np.maximum(np.abs(score_want[valid]), 1e-300)
    assert rel[score_want[valid] != 0].max() <= REL_TOL[t]


@pytest.mark.gpu
def test_photographs_at_config_d_scale_equal_the_cpu_golden_through_the_list_length_trials(engine):
    """Config PD (round 6): one of the reference's SampleImages as a 4096 x 4096 target, 5000 tiles of 64 x 64 cut from the other four,
    CIEDE2000, repeat range 4 (+500) -- config D's job on the content the tool is for; 4096 cells against the oracle's whole-grid golden
    (1001 s of CPU).  Neighbouring cells want the same few images here: the sweeps do not settle (the front form of the wavefront
    resolves), short lists do not verify (deepening by need: 4 -> 16 -> 42 entries), and over its first generates the engine TRIES list
    lengths of 4, 8 and 16 and keeps the fastest.  Every one of six consecutive generates must give the golden's grid."""
    from mosaicmagnifique_amd.engine import Engine
    e = Engine(0)                                   # a fresh engine: the trials start from nothing
    try:
        seen = []
        for call in range(6):
            tm, rel, same = whole_grid_against_golden(e, "PD")
            seen.append((tm["spec_rounds"], tm["deepened_cells"], tm["resolve_sweeps"], tm["host_syncs"], round(tm["total_ms"], 2)))
            assert tm["band_alarms"] == 0 and tm["voided_launches"] == 0 and tm["dense_cells"] == 0
        print(json.dumps(dict(config="PD", max_rel_score_err=rel, inputs_identical=same, rounds_deepened_sweeps_syncs_ms=seen)))
        assert seen[0][1] > 0.02 * 4096             # the first generate, on lists of 4, had to deepen (that is what starts the trials)
    finally:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["A", "F", "G", "H", "I", "P"])
def test_further_bench_configurations_whole_grid_equals_the_cpu_golden(engine, name):
    """bench.py's other configurations, every cell: the reference's CPU-runnable case (A), the GUI's default 128-px cells as
    RGB (F: cells too large for LDS) and as CIEDE2000 (G: the packed-fp16 pre-tier is on there by default), the harness's
    CIEDE2000 line (H: 25-px cells), 80-px cells (I: the BIG instantiation's half-group lanes), photographs with the repeat
    penalty (P: deepening rounds, resumed sweeps, learned list lengths -- run twice, so that the second call starts from what
    the first one learned)."""
    tm, rel, same = whole_grid_against_golden(engine, name)
    if name == "P":
        tm2, _, _ = whole_grid_against_golden(engine, name)
        assert tm2["spec_rounds"] <= tm["spec_rounds"]
    print(json.dumps(dict(config=name, max_rel_score_err=rel, inputs_identical=same)))
