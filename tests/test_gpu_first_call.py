"""The first-call path: what generateBestFits() of a FRESH generator costs (the reference builds one per run,
MainWindow.cpp:584-607; its harness times exactly that, Benchmark_Generator.h:83-104).  mm_destroy parks the
engine's device resources and the next mm_create on the device adopts them (include/mosaic_hip.h): nothing logical
may survive the hand-over -- a new engine has nothing staged, the default options, and gives the oracle's grid on
whatever it is given next, whatever the previous engine left in the buffers."""
import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import GridUtility as G
from mosaicmagnifique_amd.colourdifference import Type
from mosaicmagnifique_amd.engine import Engine, EngineError, trim_cache
from oracle_ffi import shape_of

pytestmark = pytest.mark.gpu


def run(oracle, e, shape, rows, cols, n_lib, t, rep=(0, 0), u8=False):
    from mosaicmagnifique_amd import synth
    size = shape.getSize()
    group = cases.group_of(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)
    if u8:
        e.set_library_u8(synth.library_u8(n_lib, size), int(t))
        e.set_target_u8(synth.target_u8(rows, cols, lattice=max(size, 8)), int(t))
        img, lib = e.get_target(rows, cols), e.get_library()
    else:
        img, lib = cases.inputs(rows, cols, size, n_lib, t)
        e.set_library(lib)
        e.set_target(img)
    e.set_masks(shape.masks4())
    best, score = e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
    want, want_score, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
    assert np.array_equal(np.where(best == n_lib, -1, best), want)
    return e.stage_timing()


def test_a_fresh_engine_adopts_the_parked_resources_and_nothing_else(oracle):
    trim_cache()
    a = Engine(0)
    first = run(oracle, a, cases.square_shape(32), 256, 320, 120, Type.CIEDE2000, rep=(2, 300), u8=True)
    assert first["reused_engine"] == 0 and first["allocations"] > 10 and first["alloc_ms"] > 0
    a.set_option("filter", 0)                     # an option and a staged library that must NOT carry over
    a.close()

    b = Engine(0)
    clocks = b.stage_timing()
    assert clocks["reused_engine"] == 1 and clocks["allocations"] == 0
    with pytest.raises(EngineError) as err:        # adopted buffers are not a staged library
        b.generate(np.zeros(0, dtype=G.CELL_DTYPE), 4, 4, 2)
    assert "MM_NOT_READY" in str(err.value)
    assert b.n_library == 0
    # the same job again: no allocation at all, the same grid; the two-tier path ran (the option did not survive)
    again = run(oracle, b, cases.square_shape(32), 256, 320, 120, Type.CIEDE2000, rep=(2, 300), u8=True)
    assert again["allocations"] == 0, again
    assert b.timing()["rescore_ms"] > 0
    b.close()

    # other shapes / sizes / types on adopted buffers: smaller cells (the tiny path's position-major copy), a hexagon
    # mask (other variants), larger everything (buffers grow), RGB
    for shape, rows, cols, n_lib, t, rep in ((cases.square_shape(8), 64, 96, 40, Type.CIEDE2000, (1, 50)),
                                            (cases.hexagon_shape(), 200, 300, 60, Type.CIEDE2000, (0, 0)),
                                            (cases.square_shape(48), 300, 400, 300, Type.RGB_EUCLIDEAN, (2, 3000)),
                                            (cases.flippy_shape(), 130, 170, 90, Type.CIE76, (0, 0))):
        c = Engine(0)
        assert c.stage_timing()["reused_engine"] == 1
        run(oracle, c, shape, rows, cols, n_lib, t, rep)
        c.close()


def test_two_live_engines_do_not_share_resources_and_trim_releases_the_parked_set(oracle):
    trim_cache()
    a, b = Engine(0), Engine(0)
    assert a.stage_timing()["reused_engine"] == 0 and b.stage_timing()["reused_engine"] == 0
    run(oracle, a, cases.square_shape(16), 96, 128, 50, Type.CIEDE2000)
    run(oracle, b, cases.square_shape(24), 96, 128, 70, Type.RGB_EUCLIDEAN)
    run(oracle, a, cases.square_shape(16), 96, 128, 50, Type.CIEDE2000)       # a's buffers were not touched by b
    a.close()                                     # parked
    b.close()                                     # one set per device: released
    assert trim_cache() == 1
    assert trim_cache() == 0
    c = Engine(0)
    assert c.stage_timing()["reused_engine"] == 0
    c.close()


def test_a_second_generator_of_a_benchmark_sized_job_allocates_nothing_under_the_default_budget(oracle):
    """Round 5 capped a parked set at 256 MB and the driver's box then paid 73-76 ms of hipMalloc per fresh config-D generator
    (BENCH_r05.json).  The cap is a budget now -- 5 % of the device memory that would be free with the set released -- so the
    workspaces of a job well above 256 MB (here 0.55 GB: 2048 x 2048 target, 1500 images of 64 x 64, CIEDE2000, repeat range 4)
    stay parked and the next generator makes NO device allocation; MOSAIC_HIP_KEEP_MB pins a fixed cap instead (64 MB: the large
    buffers are released at mm_destroy and allocated again)."""
    import os
    from mosaicmagnifique_amd import synth
    shape = cases.square_shape(64)
    rows = cols = 2048
    n_lib, t, rep = 1500, Type.CIEDE2000, (4, 500)
    lib_u8, img_u8 = synth.library_u8(n_lib, 64), synth.target_u8(rows, cols, lattice=64)
    group = cases.group_of(shape)
    state = G.getGridState(group, rows, cols)[0]
    cells = G.cell_records(group, state, rows, cols)

    def fresh():
        e = Engine(0)
        e.set_library_u8(lib_u8, int(t))
        e.set_target_u8(img_u8, int(t))
        e.set_masks(shape.masks4())
        best, _ = e.generate(cells, state.shape[1], state.shape[0], int(t), *rep)
        clocks = e.stage_timing()
        e.close()
        return best, clocks

    trim_cache()
    saved = os.environ.pop("MOSAIC_HIP_KEEP_MB", None)
    try:
        first, c1 = fresh()
        assert c1["reused_engine"] == 0 and c1["allocated_bytes"] > (400 << 20), c1      # far above round 5's fixed cap
        second, c2 = fresh()
        assert c2["reused_engine"] == 1 and c2["allocations"] == 0 and c2["alloc_ms"] == 0.0, c2
        assert np.array_equal(first, second)
        os.environ["MOSAIC_HIP_KEEP_MB"] = "64"       # (read at mm_destroy)
        third, c3 = fresh()                           # adopts the full set the second one parked; parks at most 64 MB itself
        assert c3["allocations"] == 0
        fourth, c4 = fresh()
        assert c4["reused_engine"] == 1 and c4["allocations"] > 0 and c4["allocated_bytes"] > (300 << 20), c4
        assert np.array_equal(first, fourth)
    finally:
        if saved is None:
            os.environ.pop("MOSAIC_HIP_KEEP_MB", None)
        else:
            os.environ["MOSAIC_HIP_KEEP_MB"] = saved
        trim_cache()
