"""Tier 1 of the two-tier scoring: the fp32 filter form never decides a result, but its
error against the fp64 form must stay inside the band the engine trusts it with
(engine.hip filter_band: 1e-3 + 6e-5 * value per CIEDE2000 pixel pair, 5e-4 per Euclid pair), and
every escape hatch (discontinuity log, contender overflow, canary) must hand the
cell to the fp64-only path with the result unchanged."""
import os

import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd.colourdifference import Type
from oracle_ffi import shape_of

pytestmark = pytest.mark.gpu

LO, HI = np.array([0, -128, -128]), np.array([100, 127, 127])
# the packed-fp16 pre-tier's band per pixel pair (launch.hpp kPreBandAbs, kPreBandRel: the worst-case bound of tools/filter_error_bound.py
# with a tenth of room; observed errors must stay far below it -- round 4's band, twice the worst observed pair, was 0.29 + 8e-3 v)
PRE_ABS, PRE_REL = (float(v) for v in os.environ.get("MM_TEST_PRE_BAND", "0.45,3.8e-2").split(","))
# ... and the share of it an OBSERVED over-estimate may reach (the worst pair of round 4's searches sits at 0.3 of this band)
PRE_OBSERVED_SHARE = 0.35


def lab_sets(rng, n):
    uni = lambda: rng.uniform(LO, HI, (n, 3))
    yield "uniform", uni(), uni()
    a = uni(); yield "near-identical", a, a + rng.normal(0, 0.3, (n, 3))
    a = uni(); a[:, 1:] *= 1e-3; yield "near-grey", a, uni()
    a = uni(); a[:, 1:] = 0; yield "grey-first", a, uni()
    a, b = uni(), uni(); b[:, 1:] = 0; yield "grey-second", a, b
    a, b = uni(), uni(); a[:, 1:] = 0; b[:, 1:] = 0; yield "grey-both", a, b
    # a grey colour whose zeros carry a minus sign: dot and cross come out as -0.0, which must not send the pair down the obtuse arm
    a, b = uni(), uni(); a[:, 1:] = -0.0; yield "grey-first-negative-zero", a, b
    a, b = uni(), uni(); b[:, 1] = -0.0; b[:, 2] = 0.0; yield "grey-second-mixed-zero", a, b
    th, r1, r2 = rng.uniform(0, 2 * np.pi, n), rng.uniform(1, 120, n), rng.uniform(1, 120, n)
    for spread in (1e-2, 1e-4):
        d = rng.normal(0, spread, n)
        yield ("opposite-%g" % spread,
               np.stack([rng.uniform(0, 100, n), r1 * np.cos(th), r1 * np.sin(th)], 1),
               np.stack([rng.uniform(0, 100, n), r2 * np.cos(th + np.pi + d), r2 * np.sin(th + np.pi + d)], 1))
    th = np.deg2rad(rng.uniform(230, 320, n)); th2 = th + rng.normal(0, 0.2, n)
    yield ("blue-rotation-term",
           np.stack([rng.uniform(0, 100, n), r1 * np.cos(th), r1 * np.sin(th)], 1),
           np.stack([rng.uniform(0, 100, n), r2 * np.cos(th2), r2 * np.sin(th2)], 1))


@pytest.mark.parametrize("prepared", [0, 1])
def test_ciede2000_filter_error_is_inside_the_band(engine, prepared):
    """Both forms of the fp32 filter: operands as given, and prepared as the dominant kernel's interleaved copies hold them
    (L / 2 - 25, chroma / 50; option probe_prepared)."""
    rng = np.random.default_rng(3)
    engine.set_option("probe_prepared", prepared)
    try:
        for name, a, b in lab_sets(rng, 1 << 21):
            a, b = a.astype(np.float32), b.astype(np.float32)
            exact = engine.colour_diff_batch(2, a, b)
            fast = engine.filter_diff_batch(2, a, b).astype(np.float64)
            handed_over = fast < 0                     # pairs the filter gives to the fp64 form
            err = np.abs(fast - exact)[~handed_over]
            # the engine trusts 1e-3 + 6e-5 * value per pixel pair (engine.hip filter_band);
            # hold the measured error to half of that
            slack = err - (5e-4 + 3e-5 * exact[~handed_over])
            assert slack.max() <= 0, (name, err.max())
            assert np.isfinite(fast).all(), name
            if name.startswith("opposite"):
                assert handed_over.any()
            else:
                assert handed_over.mean() < 1e-4, name
    finally:
        engine.set_option("probe_prepared", 0)


def test_ciede2000_filter_band_structured_sweep(engine):
    """What backs the band the engine trusts the filter with (engine.hip filter_band: 1e-3 + 6e-5 * value
    per pixel pair): a structured sweep generated and reduced on the device -- hue x hue (512 x 512 jittered
    steps of the circle) x chroma x chroma (24 x 24 levels from exactly grey over 1e-4 to 150) x 25 lightness
    pairs, three jitter seeds = 1.1e10 pairs -- and both sides of the opposite-hue discontinuity (hue
    difference 180 deg +- 1e-7 .. 3e-2 rad: where the fp32 branch choice could differ from the fp64 one).
    The worst pair must use less than half of the trusted band."""
    grid = 512 * 512 * 24 * 24 * 25
    measured = 0
    for seed in range(3):
        r = engine.filter_band_sweep(0, grid, seed=seed)
        print("band sweep mode 0 seed %d: %s" % (seed, r))
        measured += r["measured"]
        assert r["band_share"] <= 0.5, r
        assert r["max_value"] > 100                      # the sweep reaches the far corners of Lab
        assert r["handed_over"] < 1e-4 * grid
    assert measured >= 1.1e10
    r = engine.filter_band_sweep(1, 512 * 128 * 24 * 24 * 25, seed=7)
    print("band sweep mode 1: %s" % r)
    assert r["band_share"] <= 0.5, r
    assert r["handed_over"] > 0 and r["measured"] > 5e8    # both: pairs inside the hand-over zone and just outside it


def _has_pretier():
    try:
        from mosaicmagnifique_amd.engine import has_pretier
        return has_pretier()
    except Exception:                     # (library not built yet at collection time: only the pre-tier's own tests depend on this)
        return False


# the packed-fp16 pre-tier is compiled in only with -DMM_WITH_PRETIER=1 (round 6: out of the default build -- on its certified band it
# loses on every benchmarked configuration); its tests run against such a build (MOSAIC_HIP_LIB=gpurun_out/pretier.so)
needs_pretier = pytest.mark.skipif(not _has_pretier(), reason="libmosaic_hip.so was built without the packed-fp16 pre-tier (MM_WITH_PRETIER=0)")


@needs_pretier
def test_packed_fp16_pretier_band_structured_sweep(engine):
    """The filter's packed-fp16 pre-tier (colour_diff_fast.hpp ciede2000_h2) takes early-exit decisions only, against the
    limit widened by ITS band (launch.hpp kPreBandAbs + kPreBandRel * value per pixel pair).  What must hold is the over-estimate:
    fp16 form - fp64 form <= band on operands rounded to halfs as the engine stores them (an under-estimate only delays an
    exit; pairs near the opposite-hue seam count 0).  The same two structured sweeps that back the fp32 band -- hue x hue x
    chroma x chroma x lightness, and both sides of the seam -- : the worst OBSERVED over-estimate stays below a third of the band (which is
    the worst-case bound of tools/filter_error_bound.py, tests/test_filter_error_bound.py), and
    nothing comes out non-finite inside the box (reported as a share of 1e9)."""
    grid = 512 * 512 * 24 * 24 * 25
    for seed in range(2):
        r = engine.filter_band_sweep(2, grid, seed=seed)
        print("fp16 pre-tier sweep mode 2 seed %d: %s" % (seed, r))
        assert r["band_share"] <= PRE_OBSERVED_SHARE, r
        assert r["measured"] > 2.5e9 and r["max_value"] > 100        # (six of the grid's 24 chroma levels lie below 1 / 16: not judged, counted as handed over)
    r = engine.filter_band_sweep(3, 512 * 128 * 24 * 24 * 25, seed=7)
    print("fp16 pre-tier sweep mode 3: %s" % r)
    assert r["band_share"] <= PRE_OBSERVED_SHARE, r
    assert r["handed_over"] > 0 and r["measured"] > 3e7        # (most of this sweep lies inside the seam zone of the fp16 form, or below its chroma floor)


def band_share(engine, a, b, fp16=False):
    """|fp32 filter form - fp64 form| as a share of the band the engine trusts (engine.hip filter_band: 1e-3 + 6e-5 * value);
    pairs the filter hands to the fp64 form count as 0.  fp16: the packed-fp16 pre-tier's form against ITS band (filter.hip
    kPreBandAbs + kPreBandRel * value) -- the signed over-estimate, which is what an early exit relies on; seam pairs count as 0."""
    exact = engine.colour_diff_batch(2, a, b)
    fast = engine.filter_diff_batch(2, a, b).astype(np.float64)
    share = (fast - exact) / (PRE_ABS + PRE_REL * exact) if fp16 else np.abs(fast - exact) / (1e-3 + 6e-5 * exact)
    share[fast < 0] = 0.0
    assert np.isfinite(fast).all()
    return share, exact


def tiny_chroma_sets(rng, n):
    """Families for the packed-fp16 form: a chroma of 1e-7 .. 1e-1 (greys as a float conversion leaves them: squares that
    underflow in halfs) beside a saturated colour, beside another such grey, and channels on rounding boundaries of halfs."""
    def tiny():
        th, c = rng.uniform(0, 2 * np.pi, n), 10.0 ** rng.uniform(-7, -1, n)
        return np.stack([rng.uniform(0, 100, n), c * np.cos(th), c * np.sin(th)], 1)
    def saturated():
        th, c = rng.uniform(0, 2 * np.pi, n), rng.uniform(60, 128, n)
        return np.stack([rng.uniform(0, 100, n), c * np.cos(th), c * np.sin(th)], 1)
    yield "tiny-chroma-first", tiny(), saturated()
    yield "tiny-chroma-second", saturated(), tiny()
    yield "tiny-chroma-both", tiny(), tiny()
    a, b = rng.uniform(LO, HI, (n, 3)), rng.uniform(LO, HI, (n, 3))
    a[:, 1:] = np.round(a[:, 1:] * 8) / 8 + 1.0 / 16       # halfway between two halfs of a / 4 for |a| in 64 .. 128 (the ulp there is 1/8 x 4)
    yield "half-rounding-ties", a, b


def climb(engine, prepared, seed, generations=80, n=1 << 17):
    """The adversarial maximiser: returns (share it started from, share it reached, worst pair a, b, its fp64 value)."""
    rng = np.random.default_rng(seed)
    fp16 = prepared == 2
    lo, hi = LO.astype(np.float32), HI.astype(np.float32)
    sets = list(lab_sets(rng, n // 8)) + (list(tiny_chroma_sets(rng, n // 8)) if fp16 else [])
    fams = [(a.astype(np.float32), b.astype(np.float32)) for _, a, b in sets]
    pick = rng.permutation(len(fams))[:8]
    if fp16:                                                       # the four half-precision families always take part
        pick = np.concatenate([pick[:4], np.arange(len(fams) - 4, len(fams))])
    a = np.concatenate([fams[k][0] for k in pick])[:n]
    b = np.concatenate([fams[k][1] for k in pick])[:n]
    corners = rng.integers(0, 2, (n // 16, 3))
    a[: n // 16] = np.where(corners, hi, lo)
    a, b = np.clip(a, lo, hi), np.clip(b, lo, hi)
    engine.set_option("probe_prepared", prepared)
    try:
        best, _ = band_share(engine, a, b, fp16)
        start = best.max()
        for gen in range(generations):
            ca, cb = a.copy(), b.copy()
            scale = (10.0 ** rng.uniform(-6, 1.5, (n, 1))).astype(np.float32)
            which = rng.integers(0, 3, (n, 1))                     # 0: first operand, 1: second, 2: both
            coord = rng.uniform(size=(n, 3)) < rng.uniform(0.34, 1.0, (n, 1))
            ulps = rng.uniform(size=n) < 0.15                      # single-ulp moves (rounding boundaries of the fp32 form)
            da = (rng.normal(0, 1, (n, 3)).astype(np.float32) * scale) * coord * (which != 1)
            db = (rng.normal(0, 1, (n, 3)).astype(np.float32) * scale) * coord * (which != 0)
            if fp16:                                               # multiplicative moves as well: a chroma of 1e-5 is reached by shrinking, not by steps
                shrink = rng.uniform(size=(n, 1)) < 0.2
                f = (10.0 ** rng.uniform(-1, 1, (n, 1))).astype(np.float32)
                da = np.where(shrink & (which != 1), ca * (f - 1) * np.array([0, 1, 1], np.float32), da)
                db = np.where(shrink & (which != 0), cb * (f - 1) * np.array([0, 1, 1], np.float32), db)
            ca += da; cb += db
            step = rng.integers(-2, 3, (n, 3)).astype(np.float32)
            ca[ulps] = np.nextafter(a[ulps], a[ulps] + step[ulps] * np.float32(1e3))       # one ulp towards sign(step)
            cb[ulps] = b[ulps]
            ca, cb = np.clip(ca, lo, hi), np.clip(cb, lo, hi)
            child, _ = band_share(engine, ca, cb, fp16)
            better = child > best
            a[better], b[better], best[better] = ca[better], cb[better], child[better]
            if gen % 8 == 7:                                       # selection: the stronger half breeds
                order = np.argsort(best)
                weak, strong = order[: n // 2], order[n // 2:]
                a[weak], b[weak], best[weak] = a[strong], b[strong], best[strong]
        worst = int(np.argmax(best))
        share, exact = band_share(engine, a[worst:worst + 1], b[worst:worst + 1], fp16)
        return float(start), float(best.max()), a[worst].tolist(), b[worst].tolist(), float(exact[0])
    finally:
        engine.set_option("probe_prepared", 0)


@pytest.mark.parametrize("prepared", [0, 1, pytest.param(2, marks=needs_pretier)])
def test_ciede2000_filter_band_adversarial_maximiser(engine, prepared):
    """Random and structured sweeps find the TYPICAL worst error; this one climbs: 2^17 pairs of f32 Lab colours -- seeded
    on the families where the fp32 form is weakest (near-grey with either operand, saturated blue where R_T peaks, both
    sides of the opposite-hue seam, near-identical colours, the corners of the Lab box) -- are mutated over 80 generations
    (both operands or one, one coordinate or all, steps from 30 units down to single ulps; the better of parent and child
    survives, every eighth generation the weaker half is replaced by copies of the stronger half) to MAXIMISE
    |fp32 - fp64| / band.  The worst pair it finds must stay below half of the trusted band, in both operand forms.
    prepared = 2: the same climb on the packed-fp16 pre-tier's form (operands rounded to halfs) against the pre-tier's own
    band, maximising the OVER-estimate fp16 - fp64 (what an early exit relies on), with the families of half precision
    (tiny_chroma_sets) in the population and multiplicative moves of the chroma.  (Its first run found a violation the
    sweeps' chroma levels step over -- fixed in ciede2000_h2, profiles/r04_pretier_maximiser.txt; more seeds and
    generations: tools/experiments/archive/r4_maximiser_long.py.)"""
    start, best, a, b, exact = climb(engine, prepared, 17 + prepared)
    print("adversarial maximiser (prepared=%d): band share %.3f -> %.3f; worst pair %s vs %s, dE00 %.6f" % (prepared, start, best, a, b, exact))
    assert best >= start                                           # it climbed (or started at its summit)
    # (the fp32 band: half of it; the fp16 band is the worst-case bound itself: an observed error reaches a third of that)
    assert best < (PRE_OBSERVED_SHARE if prepared == 2 else 0.5), (a, b, best)


def test_band_guard_is_silent_on_honest_content_and_loud_on_a_broken_band(oracle, engine):
    """The run-time guard of the band: on sampled cells the best candidate the band EXCLUDED rides along
    as a sentinel and is re-scored in fp64; it must never beat a contender (mm_timing.band_alarms stays 0,
    also with every cell sampled and with hundreds of near-duplicates crowding the band's edge).  That
    the guard fires when the band is wrong is shown with a deliberately inverted one (option band_pct = -100:
    the threshold drops below the K-th lowest filter score, so true contenders are excluded)."""
    t = Type.CIEDE2000
    shape = cases.square_shape(32)
    img, lib = cases.inputs(160, 192, 32, 300, t)
    lib2 = lib.copy()
    rng = np.random.default_rng(5)
    for i in range(1, 250):
        lib2[i] = lib2[0] + rng.normal(0, 2e-4, lib2[0].shape).astype(np.float32)
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    try:
        engine.set_option("band_guard", 1)               # every cell carries a sentinel (the default since round 4)
        tm = run_both(oracle, engine, shape, img, lib, t, rep=(2, 300))
        assert tm["band_alarms"] == 0 and tm["dense_cells"] == 0
        tm = run_both(oracle, engine, shape, img, lib2, t)
        assert tm["band_alarms"] == 0
        engine.set_option("band_pct", -100)              # the band inside out: the threshold falls BELOW the K-th score
        engine.set_option("prune", 0)                    # (every score finite: the excluded contender is there to be seen)
        engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
        engine.generate(cells, state.shape[1], state.shape[0], int(t))
        assert engine.timing()["band_alarms"] == len(cells)     # every cell: a true contender was excluded, the guard saw it
    finally:
        engine.set_option("band_pct", 100)
        engine.set_option("prune", 1)
        engine.set_option("band_guard", 1)


def test_euclid_filter_error_is_inside_the_band(engine):
    rng = np.random.default_rng(4)
    a = rng.uniform(LO, HI, (1 << 21, 3)).astype(np.float32)
    b = rng.uniform(LO, HI, (1 << 21, 3)).astype(np.float32)
    err = np.abs(engine.filter_diff_batch(1, a, b) - engine.colour_diff_batch(1, a, b))
    assert err.max() <= 1e-4                                # band trusts 5e-4
    a = rng.integers(0, 256, (1 << 21, 3)).astype(np.float32)
    b = rng.integers(0, 256, (1 << 21, 3)).astype(np.float32)
    err = np.abs(engine.filter_diff_batch(0, a, b) - engine.colour_diff_batch(0, a, b))
    assert err.max() <= 1e-4


def run_both(oracle, engine, shape, img, lib, t, rep=(0, 0), masks=None):
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    masks = shape.masks4() if masks is None else masks
    engine.set_library(lib); engine.set_masks(masks); engine.set_target(img)
    engine.set_option("filter", 1)
    engine.set_option("prune", 0)
    bestn, scoren = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    engine.set_option("prune", 1)
    for lds in (0, 1):                                       # filter kernel without / with the LDS cell tile
        engine.set_option("lds_filter", lds)
        bestg, scoreg = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
        assert np.array_equal(bestg, bestn) and np.array_equal(scoreg[state >= 0], scoren[state >= 0])
    engine.set_option("lds_filter", -1)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    tm = engine.timing()
    assert np.array_equal(best, bestn) and np.array_equal(score[state >= 0], scoren[state >= 0])
    engine.set_option("filter", 0)
    best0, score0 = engine.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    engine.set_option("filter", 1)
    want, _, _ = oracle.generate_best_fits(shape_of(shape), img, lib, masks, int(t), *rep)
    assert np.array_equal(best, best0)
    assert np.array_equal(score[state >= 0], score0[state >= 0])       # bit-identical across the two paths
    assert np.array_equal(np.where(best == len(lib), -1, best), want)
    return tm


@pytest.mark.parametrize("first", [0, 1, 2, 3])
def test_four_masks_of_different_areas_keep_the_whole_image_bound_sound(oracle, engine, first):
    """The C ABI takes ANY four masks (mm_set_masks), not only the four flips of one shape: here they hold 1 / 8, 1 / 2,
    3 / 4 and all of the cell's pixels, rotated so that each of them is variant 0 once.  Stage 0 of the filter's bound
    compares whole-image statistics of a cell and of a library image over the blocks of the CELL'S variant -- computed
    over variant 0's block count for every cell (round 3) a cell of a larger variant got a partial sum on one side
    and the bound could drop the true best fit.  Pruned (stage 0 on and off) == unpruned == fp64-only == oracle."""
    size = 64
    shape = cases.flippy_shape(size)                      # all four flip states occur in its grid
    yy, xx = np.mgrid[0:size, 0:size]
    four = [np.where((xx < 16) & (yy < 32), 255, 0), np.where(xx <= yy, 255, 0),
            np.where((xx < 48) | (yy < 16), 255, 0), np.full((size, size), 255)]
    masks = np.stack([four[(k + first) % 4] for k in range(4)]).astype(np.uint8)
    assert len({int((m != 0).sum()) for m in masks}) == 4
    img, lib = cases.inputs(5 * size, 6 * size, size, 160, Type.CIEDE2000, lattice=size)
    for stage0 in (1, 0):
        engine.set_option("stage0", stage0)
        try:
            tm = run_both(oracle, engine, shape, img, lib, Type.CIEDE2000, rep=(1, 400), masks=masks)
        finally:
            engine.set_option("stage0", 1)
        assert tm["seed_ms"] > 0 and tm["band_alarms"] == 0


def opposite_hue_inputs(rng, size, n_lib, rows, cols, jitter):
    """Target of one hue, library images of (almost exactly) the opposite hue: every pixel
    pair sits on CIEDE2000's |dh| = 180 deg discontinuity."""
    th = 0.7
    img = np.empty((rows, cols, 3), np.float32)
    img[..., 0] = rng.uniform(30, 70, (rows, cols))
    r = rng.uniform(20, 60, (rows, cols))
    img[..., 1], img[..., 2] = r * np.cos(th), r * np.sin(th)
    lib = np.empty((n_lib, size, size, 3), np.float32)
    lib[..., 0] = rng.uniform(30, 70, lib.shape[:3])
    r = rng.uniform(20, 60, lib.shape[:3])
    d = rng.normal(0, jitter, lib.shape[:3])
    lib[..., 1], lib[..., 2] = r * np.cos(th + np.pi + d), r * np.sin(th + np.pi + d)
    return img, lib


def test_discontinuity_pairs_are_patched_in_fp64(oracle, engine):
    rng = np.random.default_rng(8)
    img, lib = opposite_hue_inputs(rng, 16, 90, 48, 48, jitter=3e-5)     # more images than seeds: the filter pass has work
    tm = run_both(oracle, engine, cases.square_shape(16), img, lib, Type.CIEDE2000, rep=(1, 20))
    assert tm["edge_pairs"] > 1000 and tm["dense_cells"] == 0


def test_discontinuity_log_overflow_falls_back_to_fp64(oracle, engine):
    """More discontinuity pairs than the log holds (capacity >= 1e6): the whole batch is
    re-run fp64-only."""
    rng = np.random.default_rng(9)
    img, lib = opposite_hue_inputs(rng, 32, 150, 96, 96, jitter=1e-6)     # 9 cells x 150 x 1024 = 1.4e6 pairs
    tm = run_both(oracle, engine, cases.square_shape(32), img, lib, Type.CIEDE2000)
    assert tm["edge_pairs"] > 1_000_000 and tm["dense_cells"] == 9


@pytest.mark.parametrize("broken", [pytest.param("fp16 band", marks=needs_pretier), "bound", "euclid bound"])
def test_audit_voids_a_launch_that_dropped_a_contender(engine, broken):
    """A candidate that leaves the CIEDE2000 filter without a full score -- dropped by the bound's stages 0 / 1, by the
    packed-fp16 pre-tier's walk or by the fp32 walk's exit test -- is seen by neither the pre-tier's canary (survivors only) nor
    the band guard (fully scored candidates only).  So one in `audit` of them (1024; a hash of cell and candidate) is walked once
    more by the fp32 form WITHOUT the bound, and a launch in which such a candidate ends clearly below the limit is redone
    without the filter tiers.  Silent on the honest kernel with EVERY dropped candidate audited (`audit` = 1); loud on a
    deliberately broken pre-tier band (`dbg_pre_band_pct` = -3000: the limit the fp16 walk tests against drops below the
    truth) and on a deliberately broken bound (`dbg_bound_pct` = 300: the lower bound is no lower bound any more): audited at 1
    in 2 and 1 in 8 -- this job has 38 000 candidates of which some 50 lie below the limit without being seeds (at 1 in 32 one
    or none of them is drawn); a benchmark-size launch has 2 x 10^7 candidates -- the
    launch is voided and the grid is the reference one; with the audit off the same kernel returns wrong cells (the test has
    teeth).  "euclid bound": the same for the RGB / CIE76 filter (filter_lds_kernel, audit_euclid_kernel), whose bound has two
    constants of its own."""
    t = Type.RGB_EUCLIDEAN if broken == "euclid bound" else Type.CIEDE2000
    shape = cases.square_shape(64)
    img, lib = cases.inputs(8 * 64, 8 * 64, 64, 600, t)
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    run = lambda: engine.generate(cells, state.shape[1], state.shape[0], int(t), 2, 300)
    knob, value = ("dbg_pre_band_pct", -3000) if broken == "fp16 band" else ("dbg_bound_pct", 300 if t == Type.CIEDE2000 else 150)
    try:
        engine.set_option("pretier", 0); engine.set_option("prune", 0)
        want, want_score = run()
        engine.set_option("prune", 1); engine.set_option("pretier", 1 if broken == "fp16 band" else 0)
        for every in (1024, 1):
            engine.set_option("audit", every)
            best, score = run()
            tm = engine.timing()
            assert tm["dense_cells"] == 0 and tm["band_alarms"] == 0 and tm["visited_groups"] > 0 and tm["voided_launches"] == 0, tm
            assert np.array_equal(best, want) and np.array_equal(score, want_score, equal_nan=True)
        engine.set_option(knob, value)
        for every in (2, 8) if t == Type.CIEDE2000 else (1, 2):       # (the Euclidean filter audits an eighth of the rate)
            engine.set_option("audit", every)
            best, score = run()
            tm = engine.timing()
            print("broken %s, audit 1 in %d: dense_cells %d voided_launches %d" % (broken, every, tm["dense_cells"], tm["voided_launches"]))
            # the launch was voided -- a sticky word, not a count that a few thousand trips could wrap -- and every cell re-done fp64-only
            assert tm["voided_launches"] == 1 and tm["dense_cells"] == len(cells), tm
            assert np.array_equal(best, want) and np.array_equal(score, want_score, equal_nan=True)
        engine.set_option("audit", 0)
        best, score = run()
        wrong = int((best != want).sum())
        print("broken %s, audit off: %d of %d cells wrong" % (broken, wrong, len(cells)))
        # (the RGB job's seeds hold every winner: the 170 candidates the broken bound drops below the limit are runners-up)
        assert wrong > 0 or t != Type.CIEDE2000
    finally:
        engine.set_option("pretier", -1); engine.set_option("audit", 1024); engine.set_option(knob, 100); engine.set_option("prune", 1)


@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIEDE2000])
def test_too_many_contenders_fall_back_to_fp64(oracle, engine, t):
    """60 library images that differ from each other by less than the filter can resolve:
    more contenders than the list holds -> flagged -> fp64-only path, same answer."""
    rng = np.random.default_rng(10)
    img, lib = cases.inputs(64, 64, 16, 80, t)
    base = lib[11].copy()
    for i in range(20, 80):
        lib[i] = base
        lib[i, rng.integers(16), rng.integers(16), 0] += np.float32(1e-3 * (i - 19))
    img[:16, :16] = base                                   # cell (0,0) matches the cluster
    tm = run_both(oracle, engine, cases.square_shape(16), img, lib, t)
    assert tm["dense_cells"] >= 1


@pytest.mark.parametrize("kind", ["blue", "mixed-chroma"])
def test_early_exit_bound_keeps_every_contender(oracle, engine, kind):
    """64 x 64 cells (16 groups: the chroma-augmented bound of the early exit at work) on content
    where its constants are tightest against the truth: saturated blue-violet patches (R_T at its
    largest) and a mix of grey and colourful patches.  Pruned == unpruned == fp64-only == oracle."""
    rng = np.random.default_rng(12 if kind == "blue" else 13)
    size, n_lib, rows, cols = 64, 120, 128, 192

    def patches(shape_hw, n):
        if kind == "blue":
            th = np.deg2rad(rng.uniform(250, 300, (n, 1, 1)) + rng.normal(0, 3, (n,) + shape_hw))
            r = rng.uniform(25, 110, (n, 1, 1)) + rng.normal(0, 4, (n,) + shape_hw)
        else:
            th = rng.uniform(0, 2 * np.pi, (n, 1, 1)) + rng.normal(0, 0.1, (n,) + shape_hw)
            r = np.where(rng.uniform(size=(n, 1, 1)) < 0.4, 2.0, rng.uniform(5, 90, (n, 1, 1))) + np.abs(rng.normal(0, 2, (n,) + shape_hw))
        light = rng.uniform(15, 85, (n, 1, 1)) + rng.normal(0, 3, (n,) + shape_hw)
        return np.stack([light, r * np.cos(th), r * np.sin(th)], -1).astype(np.float32)

    lib = patches((size, size), n_lib)
    tiles = patches((size, size), (rows // size) * (cols // size))
    img = tiles.reshape(rows // size, cols // size, size, size, 3).transpose(0, 2, 1, 3, 4).reshape(rows, cols, 3).copy()
    tm = run_both(oracle, engine, cases.square_shape(size), img, lib, Type.CIEDE2000, rep=(1, 300))
    assert tm["dense_cells"] == 0


@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIE76])
@pytest.mark.parametrize("size", [64, 48, 16])
def test_euclid_early_exit_bound_keeps_every_contender(oracle, engine, t, size):
    """The block-mean bound of the LDS filter kernel (Euclidean differences) at 16 / 9 / 1 groups per
    cell, clipped cells included (they take the plain partial-sum exit): pruned == unpruned ==
    fp64-only == oracle, with and without repeats."""
    rows, cols = 3 * size + size // 2, 4 * size + size // 3
    img, lib = cases.inputs(rows, cols, size, 150, t, lattice=size)
    run_both(oracle, engine, cases.square_shape(size), img, lib, t)
    run_both(oracle, engine, cases.square_shape(size), img, lib, t, rep=(2, 3000))


@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIEDE2000])
@pytest.mark.parametrize("size", [128, 96, 80, 32, 8])
def test_early_exit_bound_at_every_cell_size_class(oracle, engine, t, size):
    """The bound's prelude takes 1, 2 or 4 blocks per lane (cells up to 64, 90 and 128 px wide) and
    groups start on lane boundaries only then; 80 px (25 groups) and 96 px (36 groups) sit in the
    2- and 4-block classes with ragged last groups, 8 px cells have a single, mostly empty group.
    Pruned == unpruned == fp64-only == oracle in each class."""
    n_lib = 40 if size >= 80 else 90
    rows, cols = 2 * size + size // 2, 3 * size + size // 4
    img, lib = cases.inputs(rows, cols, size, n_lib, t, lattice=size)
    run_both(oracle, engine, cases.square_shape(size), img, lib, t, rep=(1, 500))


@pytest.mark.parametrize("size", [128, 96])
@pytest.mark.parametrize("sort_groups,check_every", [(0, 0), (5, 1), (16, 3), (64, 1), (64, 7)])
def test_visiting_order_of_large_cells_does_not_change_the_result(oracle, engine, size, sort_groups, check_every):
    """Cells of 33 .. 64 pixel groups (96 px: 36 groups, a ragged last one; 128 px: all 64 lanes of the prelude) rank their
    groups over the whole wave: plane order, the n largest bounds first, all by descending bound, the exit test after every
    group or every few -- pruned == unpruned == fp64-only == oracle each time."""
    t = Type.CIEDE2000
    rows, cols = 2 * size + size // 3, 3 * size + size // 5
    img, lib = cases.inputs(rows, cols, size, 60, t, lattice=size // 2)
    try:
        engine.set_option("sort_groups", sort_groups)
        engine.set_option("check_every", check_every)
        tm = run_both(oracle, engine, cases.square_shape(size), img, lib, t, rep=(1, 400))
        assert tm["visited_groups"] > 0 and tm["band_alarms"] == 0
    finally:
        engine.set_option("sort_groups", -1)
        engine.set_option("check_every", 0)


@pytest.mark.parametrize("size", [40, 48, 64])
@pytest.mark.parametrize("sort_groups", [0, 1, 3, 16])
def test_visiting_order_of_the_groups_does_not_change_the_result(oracle, engine, size, sort_groups):
    """A candidate's 256-pixel groups may be visited in any order (the exit test is "sum of the visited groups
    + bound of the unvisited ones > limit"): plane order, the n largest bounds first then plane order, or
    all by descending bound -- pruned == unpruned == fp64-only == oracle each time; 40 and 48 px cells have
    a ragged last group, 64 px fills all 16."""
    t = Type.CIEDE2000
    rows, cols = 3 * size + size // 3, 4 * size + size // 5
    img, lib = cases.inputs(rows, cols, size, 120, t, lattice=size)
    try:
        engine.set_option("sort_groups", sort_groups)
        tm = run_both(oracle, engine, cases.square_shape(size), img, lib, t, rep=(1, 400))
        assert tm["visited_groups"] > 0 and tm["bound_dropped"] > 0
    finally:
        engine.set_option("sort_groups", -1)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [16, 24, 32, 40, 48, 56, 64])
@pytest.mark.parametrize("two_level", [1, 0])
def test_stage_one_handles_every_group_count(oracle, engine, size, two_level):
    """Stage 1 of the filter packs 64 / groups candidates into a wave pass (1, 3, 4, 7, 9, 13 and 16 groups of
    256 positions for these cell sizes; tiles of 64 candidates leave a ragged last pass and the 150-image library
    a ragged last tile): same grid as the single-stage form, the unpruned path, the fp64-only path and the oracle."""
    t = Type.CIEDE2000
    rows, cols = 4 * size + size // 2, 5 * size
    img, lib = cases.inputs(rows, cols, size, 150, t, lattice=size)
    try:
        engine.set_option("two_level", two_level)
        tm = run_both(oracle, engine, cases.square_shape(size), img, lib, t, rep=(2, 300))
        assert tm["bound_dropped"] > 0 and tm["band_alarms"] == 0
    finally:
        engine.set_option("two_level", 1)


@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIE76])
@pytest.mark.parametrize("size", [128, 112])
def test_euclid_cells_too_large_for_lds_keep_the_bound(oracle, engine, t, size):
    """128 x 128 cells at full detail -- the reference GUI's defaults (CellShape.h:36, MainWindow.cpp:150,157) -- do not fit
    LDS (196 KB of planes): the RGB / CIE76 filter keeps its block bound, one candidate per lane, and only the survivors'
    waves read the cell from global memory (filter_lds_kernel<.., CELL_GLOBAL>).  700 images = a ragged tile; clipped
    cells at the right and bottom edges.  Pruned == unpruned == fp64-only == oracle, with and without repeats."""
    rows, cols = 2 * size + size // 2, 3 * size + size // 3
    img, lib = cases.inputs(rows, cols, size, 700, t, lattice=size)
    tm = run_both(oracle, engine, cases.square_shape(size), img, lib, t)
    assert tm["bound_dropped"] > 0 and tm["visited_groups"] < tm["total_groups"]
    run_both(oracle, engine, cases.square_shape(size), img, lib, t, rep=(1, 4000))


@pytest.mark.parametrize("t", [Type.RGB_EUCLIDEAN, Type.CIE76])
@pytest.mark.parametrize("size", [160, 256])
def test_euclid_cells_beyond_64_groups_keep_stage_one(oracle, engine, t, size):
    """Above 128 x 128 the per-wave prelude of the rest bound no longer holds a cell's blocks, but stage 1 of the RGB / CIE76
    filter (one candidate per lane over all of the cell's block sums) has no such limit: it still drops most candidates and
    the survivors walk their pixels with the plain partial-sum exit.  Pruned == unpruned == fp64-only == oracle."""
    rows, cols = 2 * size + size // 3, 2 * size + size // 2
    img, lib = cases.inputs(rows, cols, size, 150, t, lattice=size // 2)
    tm = run_both(oracle, engine, cases.square_shape(size), img, lib, t, rep=(1, 3000))
    assert tm["bound_dropped"] > 0


@pytest.mark.parametrize("size", [160, 256])
def test_cells_beyond_the_bound_s_reach(oracle, engine, size):
    """Cells of more than 64 pixel groups (over 128 x 128) are outside what the early-exit bound's prelude holds in one
    wave: they run with the plain partial-sum exit -- same grid as unpruned, fp64-only and the oracle."""
    t = Type.CIEDE2000
    rows, cols = 2 * size + size // 3, 2 * size + size // 2
    img, lib = cases.inputs(rows, cols, size, 24, t, lattice=size // 2)
    tm = run_both(oracle, engine, cases.square_shape(size), img, lib, t, rep=(1, 300))
    assert tm["band_alarms"] == 0


@pytest.mark.parametrize("t", [Type.CIE76, Type.CIEDE2000])
def test_library_larger_than_any_table_in_lds(oracle, engine, t):
    """26 000 images: the seed selection keeps no per-image table in LDS (two closest per thread stripe, then the M
    closest of those 512), so the limits and the early exit work for any library size -- still the reference's picks,
    for the Lab-Euclidean and the CIEDE2000 form."""
    size, n_lib = 8, 26000
    img, lib = cases.inputs(4 * size, 6 * size, size, n_lib, t, lattice=8)
    shape = cases.square_shape(size)
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    engine.set_option("tiny", 0)            # (8 x 8 cells would take the one-wave-per-cell path: this test is about the seeds)
    try:
        best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t), 2, 50)
    finally:
        engine.set_option("tiny", 1)
    tm = engine.timing()
    assert tm["seed_ms"] > 0.02 and tm["rescore_ms"] > 0                    # two-tier, seeds and limits in place
    tiny_best, tiny_score = engine.generate(cells, state.shape[1], state.shape[0], int(t), 2, 50)     # ... and the tiny path agrees
    assert np.array_equal(tiny_best, best) and engine.timing()["seed_ms"] == 0
    grid = np.where(best < n_lib, best, -1)
    pos = np.stack([cells["grid_y"], cells["grid_x"]], axis=1)
    picks, scores = oracle.picks_given_grid(shape_of(shape), img, lib, shape.masks4(), int(t), 2, 50, grid, pos, threads=16)
    got = best[pos[:, 0], pos[:, 1]]
    assert np.array_equal(np.where(got == n_lib, -1, got), picks)


@pytest.mark.parametrize("t,where,value", [(Type.CIEDE2000, "library", (50.0, 160.0, 20.0)), (Type.CIEDE2000, "target", (104.0, 5.0, 5.0)),
                                           (Type.RGB_EUCLIDEAN, "library", (300.0, 10.0, 10.0)), (Type.CIEDE2000, "target", (50.0, float("inf"), 0.0))])
def test_values_outside_the_validated_box_run_fp64_only(oracle, engine, t, where, value):
    """The fp32 filter's band (engine.hip filter_band) is validated on a box -- Lab: L in [0, 100], chroma <= 150; RGB / CIE76:
    |channel| <= 256 (the structured sweep, the adversarial maximiser).  The f32 setters take ANY floats: a single pixel
    outside the box (or non-finite) sends the call down the fp64-only path -- mm_timing.filter_off_range says so -- and the
    grid is the oracle's all the same; back inside the box the two-tier path runs again."""
    shape = cases.square_shape(32)
    img, lib = cases.inputs(128, 160, 32, 120, t)
    img, lib = img.copy(), lib.copy()
    finite = np.isfinite(value).all()
    if where == "library":
        lib[7, 3, 5] = value
    else:
        img[40, 50] = value
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    best, score = engine.generate(cells, state.shape[1], state.shape[0], int(t))
    tm = engine.timing()
    assert tm["filter_off_range"] == 1 and tm["rescore_ms"] == 0 and tm["seed_ms"] == 0
    # (the non-finite case too since round 6: a NaN term stays NaN in the parallel sums' form, as in ColourDifference.cpp -- the cell
    # holding the pixel gets no fit from any candidate, the oracle's -1 / the engine's sentinel)
    want, _, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t))
    assert np.array_equal(np.where(best == len(lib), -1, best), want)
    if not finite:
        assert (want == -1).sum() > (state < 0).sum()
    img2, lib2 = cases.inputs(128, 160, 32, 120, t)
    engine.set_library(lib2); engine.set_target(img2)
    engine.generate(cells, state.shape[1], state.shape[0], int(t))
    tm = engine.timing()
    assert tm["filter_off_range"] == 0 and tm["rescore_ms"] > 0


@needs_pretier
def test_the_measured_pretier_rule_belongs_to_the_observed_band(oracle, engine):
    """`pretier` = -1 on the default (worst-case) band: the pre-tier never runs -- no benchmarked configuration gains there.  With
    `pretier_band` = 1 an eligible job (cells above 32 groups) spends a warm-up launch, then one launch without and one with the
    pre-tier, and keeps one form; every call returns the same grid and scores (only the TIME of a call depends on the history)."""
    t = Type.CIEDE2000
    shape = cases.square_shape(96)                           # 36 groups: eligible by cell size
    img, lib = cases.inputs(3 * 96, 4 * 96, 96, 150, t)
    group = cases.group_of(shape)
    state = G.getGridState(group, img.shape[0], img.shape[1])[0]
    cells = G.cell_records(group, state, img.shape[0], img.shape[1])
    engine.set_library(lib); engine.set_masks(shape.masks4()); engine.set_target(img)
    run = lambda: engine.generate(cells, state.shape[1], state.shape[0], int(t), 1, 300)
    try:
        engine.set_option("pretier", -1); engine.set_option("pretier_band", 0)
        want, want_score = run()
        assert [engine.timing()["pretier"]] + [run() and engine.timing()["pretier"] for _ in range(3)] == [0, 0, 0, 0]
        engine.set_option("pretier_band", 1)                  # (resets the trial)
        seen = []
        for _ in range(5):
            best, score = run()
            seen.append(engine.timing()["pretier"])
            assert np.array_equal(best, want) and np.array_equal(score, want_score, equal_nan=True)
            assert engine.timing()["voided_launches"] == 0
        assert seen[:3] == [0, 0, 1] and seen[3] == seen[4], seen       # warm-up, without, with, then the form it kept
        ref, _, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), 1, 300)
        assert np.array_equal(np.where(want == len(lib), -1, want), ref)
    finally:
        engine.set_option("pretier", -1); engine.set_option("pretier_band", 0)
