"""Shared scenario builders for the tests: cell shapes, synthetic targets / libraries."""
import json
import os

import numpy as np

from mosaicmagnifique_amd import synth
from mosaicmagnifique_amd.cellshape import CellGroup, CellShape
from mosaicmagnifique_amd.colourdifference import Type

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def square_shape(size):
    return CellShape(size)


def hexagon_shape():
    """The 64x64 detail cell of Cells/Hexagon.mcs (tests/golden/hexagon64.json)."""
    doc = json.load(open(os.path.join(GOLDEN, "hexagon64.json")))
    mask = np.array([[255 if ch == "1" else 0 for ch in row] for row in doc["rows"]], dtype=np.uint8)
    return CellShape.from_dict(doc["params"], mask)


def flippy_shape(size=24):
    """A right-triangle mask with differing alternate spacings, both offsets and all
    four flip flags set: exercises every branch of getRectAt / getFlipStateAt and
    makes the 4 mask variants pairwise different."""
    yy, xx = np.mgrid[0:size, 0:size]
    mask = np.where(xx <= yy, 255, 0).astype(np.uint8)
    s = CellShape(mask)
    s.rowSpacing, s.alternateRowSpacing = size - 3, size - 7
    s.colSpacing, s.alternateColSpacing = size - 2, size - 5
    s.alternateRowOffset, s.alternateColOffset = 5, 3
    s.alternateColFlipHorizontal = True
    s.alternateRowFlipVertical = True
    s.alternateRowFlipHorizontal = True
    s.alternateColFlipVertical = False
    return s


_last_inputs = {}


def inputs(rows, cols, size, n_lib, diff_type, lattice=None, duplicates=()):
    """_inputs, with the latest LARGE workload kept (read-only): the full-size tests ask for
    the same one several times in a row and building config D takes ~15 s of numpy."""
    if (rows * cols + n_lib * size * size) * 12 < (64 << 20):
        return _inputs(rows, cols, size, n_lib, diff_type, lattice, duplicates)
    key = (rows, cols, size, n_lib, int(diff_type), lattice, tuple(duplicates))
    if key not in _last_inputs:
        _last_inputs.clear()
        _last_inputs[key] = _inputs(rows, cols, size, n_lib, diff_type, lattice, duplicates)
    img, lib = _last_inputs[key]
    img.setflags(write=False)
    lib.setflags(write=False)
    return img, lib


def _inputs(rows, cols, size, n_lib, diff_type, lattice=None, duplicates=()):
    """(target f32 HWC, library f32 [n][size][size][3]) after the colour conversion of
    the given difference type.  duplicates: (dst, src) pairs -- library image dst
    becomes a copy of src (exact ties)."""
    lattice = lattice or max(size, 8)
    img = synth.target_u8(rows, cols, lattice=lattice)
    lib = synth.library_u8(n_lib, size)
    for dst, src in duplicates:
        lib[dst] = lib[src]
    return synth.preprocess(img, Type(diff_type)), synth.preprocess(lib, Type(diff_type))


def group_of(shape):
    return CellGroup(shape)
