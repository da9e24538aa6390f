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
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-o", out,
           os.path.join(ROOT, "tests", "host_cpp_runner.cpp"),
           os.path.join(HOST, "HIPPhotomosaicGenerator.cpp"),
           "-L" + lib_dir, "-lmosaic_hip", "-Wl,-rpath," + lib_dir]
    subprocess.run(cmd, check=True)
    return out


def test_host_cpp_compiles(tmp_path):
    """CPU tier: the C++ host builds and links against the C ABI (no GPU call)."""
    assert os.path.exists(build_runner(str(tmp_path / "runner")))


@pytest.mark.gpu
@pytest.mark.parametrize("shape_name,t,rep", [
    ("square", Type.RGB_EUCLIDEAN, (0, 0)),
    ("hexagon", Type.CIEDE2000, (2, 300)),
    ("flippy", Type.CIE76, (1, 50)),
])
def test_host_cpp_matches_oracle(oracle, tmp_path, shape_name, t, rep):
    shape = {"square": lambda: cases.square_shape(16), "hexagon": cases.hexagon_shape,
             "flippy": cases.flippy_shape}[shape_name]()
    size = shape.getSize()
    rows, cols, n_lib = 4 * size + 5, 5 * size + 9, 40
    from mosaicmagnifique_amd import synth
    lib_u8 = synth.library_u8(n_lib, size)
    img = synth.preprocess(synth.target_u8(rows, cols, lattice=max(size, 8)), t)
    lib = synth.preprocess(lib_u8, t)
    header = np.array([rows, cols, size, n_lib, int(t), rep[0], rep[1],
                       shape.rowSpacing, shape.colSpacing, shape.alternateRowSpacing, shape.alternateColSpacing,
                       shape.alternateRowOffset, shape.alternateColOffset,
                       shape.alternateColFlipHorizontal, shape.alternateColFlipVertical,
                       shape.alternateRowFlipHorizontal, shape.alternateRowFlipVertical], dtype=np.int32)
    scenario, result = tmp_path / "scenario.bin", tmp_path / "result.bin"
    with open(scenario, "wb") as f:
        f.write(header.tobytes()); f.write(img.tobytes()); f.write(lib.tobytes())
        f.write(shape.getCellMask(False, False).tobytes())
        f.write(lib_u8.tobytes())
    runner = build_runner(str(tmp_path / "runner"))
    subprocess.run([runner, str(scenario), str(result)], check=True, timeout=300)
    raw = np.fromfile(result, dtype=np.uint8)
    gh, gw = np.frombuffer(raw[:8], dtype=np.int32)
    got = np.frombuffer(raw[8:8 + 8 * gh * gw], dtype=np.int64).reshape(gh, gw)
    want, _, _ = oracle.generate_best_fits(shape_of(shape), img, lib, shape.masks4(), int(t), *rep)
    assert np.array_equal(got, want)
    mosaic = raw[8 + 8 * gh * gw:].reshape(rows, cols, 4)
    want_mosaic = oracle.build_photomosaic(shape_of(shape), rows, cols, lib_u8, shape.masks4(), want, (7, 8, 9, 10))
    assert np.array_equal(mosaic, want_mosaic)
