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


def inputs(rows, cols, size, n_lib, diff_type, lattice=None, duplicates=()):
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
