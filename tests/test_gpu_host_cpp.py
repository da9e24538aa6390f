"""The C++ host mirror (mosaicmagnifique_amd/csrc/host/HIPPhotomosaicGenerator.{h,cpp}):
compiled with g++ against libmosaic_hip.so and driven the way the reference's generator
fixtures drive theirs; the grid it returns must equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

import cases
from mosaicmagnifique_amd import _build
from mosaicmagnifique_amd.colourdifference import Type
from oracle_ffi import shape_of

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HOST = os.path.join(ROOT, "mosaicmagnifique_amd", "csrc", "host")


def build_runner(out):
    lib_dir = os.path.dirname(_build.build())
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-Wextra", "-o", out,
           os.path.join(ROOT, "tests", "host_cpp_runner.cpp"),
           os.path.join(HOST, "HIPPhotomosaicGenerator.cpp"),
           "-L" + lib_dir, "-lmosaic_hip", "-Wl,-rpath," + lib_dir, "-lz", "-pthread"]
    subprocess.run(cmd, check=True)
    return out


def test_host_cpp_compiles(tmp_path):
    """CPU tier: the C++ host builds and links against the C ABI (no GPU call)."""
    assert os.path.exists(build_runner(str(tmp_path / "runner")))


def header_of(shape, rows, cols, n_lib, t, rep, detail=100, steps=0, u8=False, scheme=0):
    return np.array([rows, cols, shape.getSize(), n_lib, int(t), rep[0], rep[1],
                     shape.rowSpacing, shape.colSpacing, shape.alternateRowSpacing, shape.alternateColSpacing,
                     shape.alternateRowOffset, shape.alternateColOffset,
                     shape.alternateColFlipHorizontal, shape.alternateColFlipVertical,
                     shape.alternateRowFlipHorizontal, shape.alternateRowFlipVertical,
                     detail, steps, int(u8) | (int(scheme) << 1)], dtype=np.int32)


def read_result(path, rows, cols):
    raw = np.fromfile(path, dtype=np.uint8)
    steps = int(np.frombuffer(raw[:4], dtype=np.int32)[0])
    pos, grids = 4, []
    for _ in range(steps):
        gh, gw = (int(v) for v in np.frombuffer(raw[pos:pos + 8], dtype=np.int32))
        pos += 8
        grids.append(np.frombuffer(raw[pos:pos + 8 * gh * gw], dtype=np.int64).reshape(gh, gw))
        pos += 8 * gh * gw
    return grids, raw[pos:].reshape(rows, cols, 4)


SHAPES = {"square": lambda: cases.square_shape(16), "hexagon": cases.hexagon_shape, "flippy": cases.flippy_shape}


@pytest.mark.gpu
@pytest.mark.parametrize("shape_name,t,rep", [
    ("square", Type.RGB_EUCLIDEAN, (0, 0)),
    ("hexagon", Type.CIEDE2000, (2, 300)),
    ("flippy", Type.CIE76, (1, 50)),
])
def test_host_cpp_matches_oracle(oracle, tmp_path, shape_name, t, rep):
    shape = SHAPES[shape_name]()
    size = shape.getSize()
    rows, cols, n_lib = 4 * size + 5, 5 * size + 9, 40
    from mosaicmagnifique_amd import synth
    lib_u8 = synth.library_u8(n_lib, size)
    img = synth.preprocess(synth.target_u8(rows, cols, lattice=max(size, 8)), t)
    lib = synth.preprocess(lib_u8, t)
    scenario, result = tmp_path / "scenario.bin", tmp_path / "result.bin"
    with open(scenario, "wb") as f:
        f.write(header_of(shape, rows, cols, n_lib, t, rep).tobytes()); f.write(img.tobytes()); f.write(lib.tobytes())
        f.write(shape.getCellMask(False, False).tobytes())
        f.write(lib_u8.tobytes())
    runner = build_runner(str(tmp_path / "runner"))
    subprocess.run([runner, str(scenario), str(result)], check=True, timeout=300)
    (got,), mosaic = read_result(result, rows, cols)
    want, _, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
    assert np.array_equal(got, want)
    want_mosaic = oracle.build_photomosaic(shape_of(shape), rows, cols, lib_u8, shape.masks4(), want, (7, 8, 9, 10))
    assert np.array_equal(mosaic, want_mosaic)


@pytest.mark.gpu
@pytest.mark.parametrize("shape_name,t,rep,detail,steps", [
    ("square32", Type.CIE76, (2, 300), 50, 1),          # CreateCellGroup(size, 1, 50), tst_CUDAGenerator.h
    ("hexagon", Type.CIEDE2000, (1, 200), 50, 2),
    ("square32", Type.RGB_EUCLIDEAN, (0, 0), 25, 0),
    ("square32", Type.CIE76, (1, 100), 75, 1),          # ratio 4/3: the general INTER_AREA of cells and library
    ("hexagon", Type.CIEDE2000, (0, 0), 40, 0),          # 64 -> 25
])
def test_host_cpp_u8_inputs_detail_and_size_steps(oracle, tmp_path, shape_name, t, rep, detail, steps):
    """The C++ host fed like the reference's fixtures (u8 BGR image and library, detail 50 %, size
    steps) against the Python host run the same way -- which test_gpu_detail_steps.py holds to the
    oracle -- grid for grid and pixel for pixel."""
    from mosaicmagnifique_amd import GridUtility as G
    from mosaicmagnifique_amd import HIPPhotomosaicGenerator, synth
    from mosaicmagnifique_amd.cellshape import CellGroup
    shape = cases.square_shape(32) if shape_name == "square32" else SHAPES[shape_name]()
    size = shape.getSize()
    rows, cols, n_lib = 5 * size + 7, 6 * size + 3, 60
    img = synth.target_u8(rows, cols, lattice=2 * size)
    rng = np.random.default_rng(11)
    img[:, cols // 2:] = rng.integers(0, 256, (rows, cols - cols // 2, 3), dtype=np.uint8)     # busy half: splits
    lib = synth.library_u8(n_lib, size)
    scenario, result = tmp_path / "scenario.bin", tmp_path / "result.bin"
    with open(scenario, "wb") as f:
        f.write(header_of(shape, rows, cols, n_lib, t, rep, detail, steps, u8=True).tobytes())
        f.write(img.tobytes()); f.write(lib.tobytes()); f.write(shape.getCellMask(False, False).tobytes())
    runner = build_runner(str(tmp_path / "runner"))
    subprocess.run([runner, str(scenario), str(result)], check=True, timeout=300)
    grids, mosaic = read_result(result, rows, cols)

    group = CellGroup(shape)
    group.setSizeSteps(steps)
    group.setDetail(detail)
    state = G.getGridState(group, rows, cols, mainImage=img)
    gen = HIPPhotomosaicGenerator(0)
    gen.setMainImage(img); gen.setLibrary(lib); gen.setColourDifference(t); gen.setCellGroup(group)
    gen.setGridState(state); gen.setRepeat(*rep)
    assert gen.generateBestFits(), gen.lastError
    want = gen.getBestFits()
    assert len(grids) == len(want) and (steps == 0 or len(want) > 1)
    for a, b in zip(grids, want):
        assert np.array_equal(a, b)
    assert np.array_equal(mosaic, gen.buildPhotomosaic((7, 8, 9, 10)))


@pytest.mark.gpu
def test_host_cpp_colour_scheme(oracle, tmp_path):
    """setColourScheme through the C++ host: the grid equals the oracle's on the scheme's last
    variant (the reference's variant cells all alias that one, PhotomosaicGeneratorBase.cpp:310)."""
    from mosaicmagnifique_amd import synth
    from mosaicmagnifique_amd.colourscheme import Type as Scheme
    shape, t, rep = cases.square_shape(16), Type.CIEDE2000, (1, 150)
    size, rows, cols, n_lib = 16, 70, 90, 50
    img = synth.target_u8(rows, cols, lattice=16)
    lib = synth.library_u8(n_lib, size)
    scenario, result = tmp_path / "scenario.bin", tmp_path / "result.bin"
    with open(scenario, "wb") as f:
        f.write(header_of(shape, rows, cols, n_lib, t, rep, u8=True, scheme=Scheme.TRIADIC).tobytes())
        f.write(img.tobytes()); f.write(lib.tobytes()); f.write(shape.getCellMask(False, False).tobytes())
    runner = build_runner(str(tmp_path / "runner"))
    subprocess.run([runner, str(scenario), str(result)], check=True, timeout=300)
    (got,), _ = read_result(result, rows, cols)
    last = oracle.hue_rotate_u8(img, 240.0)
    want, _, _ = oracle.generate_best_fits(shape_of(shape), synth.preprocess(last, t), synth.preprocess(lib, t),
                                           shape.masks4(), int(t), *rep)
    assert np.array_equal(got, want)
    plain, _, _ = oracle.generate_best_fits(shape_of(shape), synth.preprocess(img, t), synth.preprocess(lib, t),
                                            shape.masks4(), int(t), *rep)
    assert not np.array_equal(got, plain)          # the scheme really changed the picture


@pytest.mark.gpu
@pytest.mark.parametrize("devices,spec_k", [("0,0", None), ("0,0,0", "2"), ("0,0,0,0,0,0,0,0", "2")])
def test_host_cpp_several_engines_in_one_process(oracle, tmp_path, devices, spec_k):
    """The single-process multi-GPU path of the C++ host (HIPPhotomosaicGenerator(devices)): one engine
    per listed device -- two / three / eight engines on device 0 here, as one GPU is what the box has --, through
    the library's mm_group: contiguous shards, ONE gather of the candidate lists (through host memory, since RCCL
    refuses a device listed twice; ncclAllGather on distinct devices), the same verifying resolve on every engine.  The grid must equal the oracle's sequential run; spec_k = 2 forces
    the deepening rounds (short lists that the repeat penalty outruns) through the sharded path too."""
    from mosaicmagnifique_amd import synth
    shape, t, rep = cases.hexagon_shape(), Type.CIEDE2000, (2, 300)
    size = shape.getSize()
    rows, cols, n_lib = 5 * size + 5, 6 * size + 9, 60
    lib_u8 = synth.library_u8(n_lib, size)
    img = synth.preprocess(synth.target_u8(rows, cols, lattice=4 * size), t)        # smooth: neighbours collide
    lib = synth.preprocess(lib_u8, t)
    scenario, result = tmp_path / "scenario.bin", tmp_path / "result.bin"
    with open(scenario, "wb") as f:
        f.write(header_of(shape, rows, cols, n_lib, t, rep).tobytes()); f.write(img.tobytes()); f.write(lib.tobytes())
        f.write(shape.getCellMask(False, False).tobytes())
        f.write(lib_u8.tobytes())
    runner = build_runner(str(tmp_path / "runner"))
    env = dict(os.environ, MOSAIC_HOST_DEVICES=devices)
    if spec_k:
        env["MOSAIC_HOST_SPEC_K"] = spec_k
    subprocess.run([runner, str(scenario), str(result)], check=True, timeout=300, env=env)
    (got,), mosaic = read_result(result, rows, cols)
    want, _, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
    assert np.array_equal(got, want)
    want_mosaic = oracle.build_photomosaic(shape_of(shape), rows, cols, lib_u8, shape.masks4(), want, (7, 8, 9, 10))
    assert np.array_equal(mosaic, want_mosaic)
