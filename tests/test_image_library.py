"""`.mil` round trip without Qt / OpenCV -- the shape of the reference's
ImageLibrary save/load equality test (test/tst_ImageLibrary.h:13-83)."""
import struct

import numpy as np
import pytest

from mosaicmagnifique_amd import synth
from mosaicmagnifique_amd.imagelibrary import MIL_MAGIC, ImageLibrary


def test_mil_round_trip(tmp_path):
    lib = ImageLibrary(32)
    images = synth.library_u8(9, 32)
    for i, im in enumerate(images):
        lib.addImage(im, "image-%d é.jpg" % i)
    path = tmp_path / "lib.mil"
    lib.saveToFile(path)
    back = ImageLibrary.loadFromFile(path)
    assert back.getImageSize() == 32 and back.getNumberOfLibraryImages() == 9
    assert back.getNames() == lib.getNames()
    assert np.array_equal(back.getImages(), images)              # PNG is lossless; BGR order preserved
    head = open(path, "rb").read(16)
    assert struct.unpack(">IIII", head) == (MIL_MAGIC, 6, 32, 9)


def test_raw_version_5_and_bad_headers(tmp_path):
    images = synth.library_u8(2, 8)
    path = tmp_path / "old.mil"
    with open(path, "wb") as f:                                   # version 5: raw mats (CustomQDataStream.h:27-37)
        f.write(struct.pack(">IIII", MIL_MAGIC, 5, 8, 2))
        for im in images:
            f.write(struct.pack(">IIII", 16, 8, 8, im.nbytes))   # CV_8UC3 = 16
            f.write(im.tobytes())
            f.write(struct.pack(">I", 0xFFFFFFFF))                # null QString
    back = ImageLibrary.loadFromFile(path)
    assert np.array_equal(back.getImages(), images) and back.getNames() == ["", ""]

    bad = tmp_path / "bad.mil"
    bad.write_bytes(struct.pack(">IIII", 0x12345678, 6, 8, 0))
    with pytest.raises(ValueError, match="not a valid .mil"):
        ImageLibrary.loadFromFile(bad)
    bad.write_bytes(struct.pack(">IIII", MIL_MAGIC, 7, 8, 0))
    with pytest.raises(ValueError, match="newer file version"):
        ImageLibrary.loadFromFile(bad)
    bad.write_bytes(struct.pack(">IIII", MIL_MAGIC, 3, 8, 0))
    with pytest.raises(ValueError, match="outdated file version"):
        ImageLibrary.loadFromFile(bad)
    with pytest.raises(ValueError):
        ImageLibrary(8).addImage(np.zeros((4, 4, 3), np.uint8))


def test_mcs_round_trip(tmp_path):
    """`.mcs` save -> load equality (the shape of test/tst_CellShape.h:13-78)."""
    import cases
    from mosaicmagnifique_amd.cellshape import CellShape
    for shape in (cases.hexagon_shape(), cases.flippy_shape(), cases.square_shape(32)):
        path = tmp_path / "shape.mcs"
        shape.saveToFile(path)
        back = CellShape.loadFromFile(path)
        assert back.to_dict() == shape.to_dict()
        assert np.array_equal(back.masks4(), shape.masks4())


# ---- the reference's shipped cell shapes (Cells/*.mcs) and the C++ reader ----------------------------
import json  # noqa: E402
import os  # noqa: E402
import subprocess  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
CELLS_DIR = "/root/reference/Cells"
SHIPPED = json.load(open(os.path.join(HERE, "golden", "cells_mcs.json")))["shapes"]


def fnv_of(mask):
    h = 1469598103934665603
    for on in (np.asarray(mask).reshape(-1) > 127).astype(np.uint8).tolist():
        h = ((h ^ on) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def build_mcs_runner(out):
    from mosaicmagnifique_amd import _build
    lib_dir = os.path.dirname(_build.build())
    host = os.path.join(ROOT, "mosaicmagnifique_amd", "csrc", "host")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-o", out,
                    os.path.join(HERE, "mcs_reader_runner.cpp"), os.path.join(host, "HIPPhotomosaicGenerator.cpp"),
                    "-L" + lib_dir, "-lmosaic_hip", "-Wl,-rpath," + lib_dir, "-lz", "-pthread"], check=True)
    return out


def run_mcs_runner(runner, paths):
    res = subprocess.run([runner] + [str(p) for p in paths], check=True, capture_output=True, text=True, timeout=120)
    return [json.loads(line) for line in res.stdout.splitlines()]


def test_fixture_lists_all_ten_shipped_shapes():
    assert len(SHIPPED) == 10 and {s["file"] for s in SHIPPED} >= {"Hexagon.mcs", "YinAndYang.mcs", "Puzzle.mcs"}
    for s in SHIPPED:
        assert s["version"] == 8 and s["size"] == 512 and 0 < s["active"] < 512 * 512
    hexagon = next(s for s in SHIPPED if s["file"] == "Hexagon.mcs")
    # the facts SURVEY 8c lists for the shape the reference's own generator tests load (tst_CUDAGenerator.h:36)
    assert (hexagon["name"], hexagon["active"], hexagon["params"]["rowSpacing"], hexagon["params"]["colSpacing"],
            hexagon["params"]["alternateRowOffset"]) == ("Hexagon", 169400, 385, 440, 220)
    assert hexagon["resized64"]["active"] == 2648 and hexagon["resized64"]["params"]["rowSpacing"] == 48


@pytest.mark.skipif(not os.path.isdir(CELLS_DIR), reason="the reference checkout (Cells/*.mcs) is not on this machine")
def test_python_reader_on_the_shipped_shapes():
    """Every shape the reference ships, parsed by the package's reader, against the committed facts."""
    import hashlib
    from mosaicmagnifique_amd.cellshape import CellShape
    for want in SHIPPED:
        path = os.path.join(CELLS_DIR, want["file"])
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == want["file_sha256"]
        shape = CellShape.loadFromFile(path)
        mask = shape.getCellMask()
        assert (shape.name, shape.getSize(), int((mask > 127).sum())) == (want["name"], want["size"], want["active"])
        assert hashlib.sha256((mask > 127).astype("u1").tobytes()).hexdigest() == want["mask_sha256"]
        assert {k: v for k, v in shape.to_dict().items() if k != "name"} == want["params"]
        small = shape.resized(64)
        assert int((small.getCellMask() != 0).sum()) == want["resized64"]["active"]
        assert {k: v for k, v in small.to_dict().items() if k != "name"} == want["resized64"]["params"]
        for fh in (False, True):
            for fv in (False, True):
                assert np.array_equal(shape.getCellMask(fh, fv), mask[::-1 if fv else 1, ::-1 if fh else 1])


def test_cpp_reader_agrees_with_the_python_reader(tmp_path):
    """CellShape::loadFromFile of the C++ host (big-endian stream + its own PNG decoder over zlib): the
    shipped shapes when the checkout is here, and always files written by the package's writer (v8,
    PNG-encoded) and a hand-packed v7 file (raw mask, CustomQDataStream.h:27-37)."""
    import cases
    from mosaicmagnifique_amd.cellshape import MCS_MAGIC, CellShape
    runner = build_mcs_runner(str(tmp_path / "mcs_runner"))
    paths = []
    for k, shape in enumerate((cases.hexagon_shape(), cases.flippy_shape(), cases.square_shape(48))):
        shape.name = "shape-%d é" % k
        p = tmp_path / ("written-%d.mcs" % k)
        shape.saveToFile(p)
        paths.append(p)
    rng = np.random.default_rng(8)
    raw_mask = (rng.integers(0, 2, (40, 40)) * 255).astype(np.uint8)
    p = tmp_path / "v7.mcs"
    with open(p, "wb") as f:
        name = "raw seven".encode("utf-16-be")
        f.write(struct.pack(">II", MCS_MAGIC, 7)); f.write(struct.pack(">I", len(name))); f.write(name)
        f.write(struct.pack(">IIII", 0, 40, 40, raw_mask.size)); f.write(raw_mask.tobytes())
        f.write(struct.pack(">iiiiii", 31, 37, 29, 35, 4, 6)); f.write(struct.pack(">BBBB", 1, 0, 0, 1))
    paths.append(p)
    if os.path.isdir(CELLS_DIR):
        paths += [os.path.join(CELLS_DIR, s["file"]) for s in SHIPPED]
    got = run_mcs_runner(runner, paths)
    assert len(got) == len(paths)
    for path, g in zip(paths, got):
        assert "error" not in g, (path, g)
        shape = CellShape.loadFromFile(path)
        mask = shape.getCellMask()
        assert (g["name"], g["size"], g["active"]) == (shape.name, shape.getSize(), int((mask > 127).sum())), path
        assert g["mask_fnv"] == fnv_of(mask) and g["flips_ok"]
        assert g["params"] == {k: v for k, v in shape.to_dict().items() if k != "name"}
        if shape.getSize() >= 64:
            small = shape.resized(64)
            assert g["resized64"]["active"] == int((small.getCellMask() != 0).sum())
            assert g["resized64"]["params"] == {k: v for k, v in small.to_dict().items() if k != "name"}
    bad = tmp_path / "bad.mcs"
    bad.write_bytes(struct.pack(">II", 0x12345678, 8))
    assert "not a valid .mcs" in run_mcs_runner(runner, [bad])[0]["error"]


def fnv_bytes(data):
    h = 1469598103934665603
    for v in bytes(data):
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def test_cpp_mil_reader_agrees_with_the_python_reader(tmp_path):
    """ImageLibraryFile::loadFromFile of the C++ host (ImageLibrary.cpp:154-231 without Qt / OpenCV): a v6 file written by the
    package's writer (PNG-encoded, RGB order inside the PNG), a hand-packed v5 file (raw BGR mats), a v6 file whose PNGs are
    grey and RGBA, and the refusals the reference makes (magic, version) plus the ones a damaged file needs."""
    import io
    from PIL import Image
    runner = build_mcs_runner(str(tmp_path / "mcs_runner"))
    images = synth.library_u8(7, 24)
    lib = ImageLibrary(24)
    for i, im in enumerate(images):
        lib.addImage(im, "photo %d é ß.png" % i)
    v6 = tmp_path / "v6.mil"
    lib.saveToFile(v6)
    v5 = tmp_path / "v5.mil"
    with open(v5, "wb") as f:
        f.write(struct.pack(">IIII", MIL_MAGIC, 5, 24, 3))
        for im in images[:3]:
            f.write(struct.pack(">IIII", 16, 24, 24, im.nbytes)); f.write(im.tobytes())
            f.write(struct.pack(">I", 0xFFFFFFFF))
    odd = tmp_path / "odd.mil"                                      # a grey PNG and an RGBA PNG inside a v6 container
    grey = images[0][..., 1]
    with open(odd, "wb") as f:
        f.write(struct.pack(">IIII", MIL_MAGIC, 6, 24, 2))
        for pil in (Image.fromarray(grey), Image.fromarray(np.dstack([images[1][..., ::-1], np.full((24, 24), 200, np.uint8)]), "RGBA")):
            png = io.BytesIO(); pil.save(png, format="PNG"); blob = png.getvalue()
            f.write(struct.pack(">I", len(blob))); f.write(blob)
            name = "x".encode("utf-16-be"); f.write(struct.pack(">I", len(name))); f.write(name)
    got = run_mcs_runner(runner, [v6, v5, odd])
    assert got[0] == {"imageSize": 24, "count": 7, "bytes": images.nbytes, "fnv": fnv_bytes(images.tobytes()), "names": lib.getNames()}
    assert got[1] == {"imageSize": 24, "count": 3, "bytes": images[:3].nbytes, "fnv": fnv_bytes(images[:3].tobytes()), "names": ["", "", ""]}
    want_odd = np.stack([np.dstack([grey] * 3), images[1]])
    assert got[2]["count"] == 2 and got[2]["fnv"] == fnv_bytes(want_odd.tobytes()) and got[2]["names"] == ["x", "x"]
    back = ImageLibrary.loadFromFile(v6)                          # the Python reader on the same file
    assert np.array_equal(back.getImages(), images)
    bad = tmp_path / "bad.mil"
    for head, text in ((struct.pack(">IIII", 0x12345678, 6, 8, 0), "not a valid .mil"), (struct.pack(">IIII", MIL_MAGIC, 7, 8, 0), "newer file version"),
                       (struct.pack(">IIII", MIL_MAGIC, 3, 8, 0), "outdated file version"), (struct.pack(">IIII", MIL_MAGIC, 6, 0, 0), "image size"),
                       (struct.pack(">IIII", MIL_MAGIC, 6, 8, 4000000000), "count exceeds")):
        bad.write_bytes(head)
        assert text in run_mcs_runner(runner, [bad])[0]["error"]
    wrong = tmp_path / "wrong.mil"                                 # an image of another size than the header's
    with open(wrong, "wb") as f:
        f.write(struct.pack(">IIII", MIL_MAGIC, 5, 16, 1))
        f.write(struct.pack(">IIII", 16, 24, 24, images[0].nbytes)); f.write(images[0].tobytes()); f.write(struct.pack(">I", 0xFFFFFFFF))
    assert "is not 16 pixels a side" in run_mcs_runner(runner, [wrong])[0]["error"]


def test_bench_consumes_mcs_and_mil_files(tmp_path):
    """bench.py --mcs / --mil (the purpose of N4: running the benchmark on the project's own files): the
    workload built from a written .mcs and .mil carries the file's shape, resized to the configuration's
    cell size, and the file's images."""
    import sys
    import cases
    sys.path.insert(0, ROOT)
    import bench
    shape = cases.hexagon_shape()
    shape.name = "Hexagon"
    mcs, mil = tmp_path / "h.mcs", tmp_path / "l.mil"
    shape.saveToFile(mcs)
    lib = ImageLibrary(128)
    images = synth.library_u8(7, 128)
    for i, im in enumerate(images):
        lib.addImage(im, "i%d" % i)
    lib.saveToFile(mil)
    w = bench.build_workload("C", 1, mcs=str(mcs), mil=str(mil))
    assert w["L"] == 7 and w["lib_u8"].shape == (7, 64, 64, 3)          # 128 px images shrunk (INTER_AREA) to the 64 px cell
    assert np.array_equal(w["lib_u8"][3], images[3].reshape(64, 2, 64, 2, 3).astype(np.uint32).sum((1, 3)).__add__(2).__floordiv__(4).astype(np.uint8))
    assert np.array_equal(w["shape"].masks4(), shape.masks4()) and "h.mcs" in w["shape_name"]
    assert len(w["cells"]) == 1694 and bench.algorithmic_bytes(w, w["cells_single"])[1] == 4206946


def test_cpp_reader_survives_truncated_and_corrupted_files(tmp_path):
    """The C++ .mcs reader parses files from disk: every truncation and single-byte corruption of a valid file
    must end in a reported error or a valid shape -- never a crash -- under AddressSanitizer + UBSan (CPU build)."""
    import random
    import cases
    from mosaicmagnifique_amd import _build
    lib_dir = os.path.dirname(_build.build())
    host = os.path.join(ROOT, "mosaicmagnifique_amd", "csrc", "host")
    runner = str(tmp_path / "mcs_asan")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", runner,
                    os.path.join(HERE, "mcs_reader_runner.cpp"), os.path.join(host, "HIPPhotomosaicGenerator.cpp"),
                    "-L" + lib_dir, "-lmosaic_hip", "-Wl,-rpath," + lib_dir, "-lz", "-pthread"], check=True)
    shape = cases.hexagon_shape()
    shape.name = "Hexagon"
    good = tmp_path / "good.mcs"
    shape.saveToFile(good)
    src = good.read_bytes()
    paths = []
    for n in list(range(0, len(src), 37)) + [len(src) - 1]:
        p = tmp_path / ("t%d.mcs" % n)
        p.write_bytes(src[:n])
        paths.append(p)
    rnd = random.Random(1)
    for k in range(60):
        b = bytearray(src)
        b[rnd.randrange(8, len(b))] ^= 0xFF
        p = tmp_path / ("c%d.mcs" % k)
        p.write_bytes(bytes(b))
        paths.append(p)
    # ... and the same for a .mil (ImageLibraryFile::loadFromFile: PNG blobs of RGB images, QString names)
    from mosaicmagnifique_amd import synth
    from mosaicmagnifique_amd.imagelibrary import ImageLibrary
    lib = ImageLibrary(16)
    for i, im in enumerate(synth.library_u8(3, 16)):
        lib.addImage(im, "n%d" % i)
    good_mil = tmp_path / "good.mil"
    lib.saveToFile(good_mil)
    src_mil = good_mil.read_bytes()
    n_mcs = len(paths)
    for n in list(range(0, len(src_mil), 53)) + [len(src_mil) - 1]:
        p = tmp_path / ("t%d.mil" % n)
        p.write_bytes(src_mil[:n])
        paths.append(p)
    n_trunc_mil = len(paths) - n_mcs
    for k in range(80):
        b = bytearray(src_mil)
        b[rnd.randrange(4, len(b))] ^= rnd.choice((0xFF, 0x01, 0x80))
        p = tmp_path / ("c%d.mil" % k)
        p.write_bytes(bytes(b))
        paths.append(p)
    res = subprocess.run([runner] + [str(p) for p in paths], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert res.returncode == 0 and "ERROR" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-2000:]
    out = [json.loads(line) for line in res.stdout.splitlines()]
    assert len(out) == len(paths)
    assert sum("error" in o for o in out[:len(src) // 37 + 2]) == len(src) // 37 + 2        # every truncation is refused
    assert all("error" in o for o in out[n_mcs:n_mcs + n_trunc_mil])                          # ... of the .mil as well
