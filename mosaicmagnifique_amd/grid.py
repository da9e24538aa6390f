"""Grid geometry of the best-fit path (host side).

Mirror of ``GridUtility`` (src/Grid/GridUtility.{h,cpp}) and of the part of
``GridGenerator::getGridState`` that applies with size steps 0
(src/Grid/GridGenerator.cpp:29-110,113-140,191-192), plus the marshalling of
valid cells into the ``mm_cell`` records the C ABI takes.

A ``StepBestFit`` is a 2-D ``int64`` array over the PADDED grid; ``NULLOPT``
(-1) stands for ``std::nullopt``, 0 for the initial ``optional(0)``, and after
generation each valid entry holds the chosen library index
(GridUtility.h:29-31).
"""
from collections import namedtuple

import numpy as np

PAD_GRID = 2          # GridUtility.h:33
NULLOPT = -1

Rect = namedtuple("Rect", "x y width height")
FlipState = namedtuple("FlipState", "horizontal vertical")

# Field order of ``struct mm_cell`` in include/mosaic_hip.h.
CELL_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"),
                       ("row_min", "<i4"), ("row_max", "<i4"),
                       ("col_min", "<i4"), ("col_max", "<i4"),
                       ("mask_id", "<i4"), ("grid_x", "<i4"), ("grid_y", "<i4")])


def _tdiv2(n):
    """C++ ``n / 2`` (truncation toward zero)."""
    return int(n / 2) if not isinstance(n, np.ndarray) else np.trunc(n / 2).astype(np.int64)


def _odd(n):
    """C++ ``n % 2 != 0`` (true for negative odd n as well)."""
    return (n % 2) != 0


def calculateGridSize(cellShape, imageWidth, imageHeight, pad=PAD_GRID):
    """GridUtility::calculateGridSize (GridUtility.cpp:25-52) -> (gridW, gridH)."""
    def along(extent, spacing, alt):
        if spacing != alt:
            return 2 * ((extent + spacing + alt - 1) // (spacing + alt))
        return (extent + spacing - 1) // spacing
    return (along(imageWidth, cellShape.colSpacing, cellShape.alternateColSpacing) + pad,
            along(imageHeight, cellShape.rowSpacing, cellShape.alternateRowSpacing) + pad)


def _axis_start(n, spacing, alt):
    half = _tdiv2(n)
    rest = n - half
    neg = rest * spacing + half * alt          # GridUtility.cpp:95-97 (roles swap for n < 0)
    pos = half * spacing + rest * alt
    return np.where(n < 0, neg, pos) if isinstance(n, np.ndarray) else (neg if n < 0 else pos)


def getRectAt(cellShape, x, y):
    """GridUtility::getRectAt (GridUtility.cpp:86-115).  x, y: ints or arrays."""
    rx = _axis_start(x, cellShape.colSpacing, cellShape.alternateColSpacing)
    ry = _axis_start(y, cellShape.rowSpacing, cellShape.alternateRowSpacing)
    if isinstance(x, np.ndarray):
        rx = rx + np.where(_odd(y), cellShape.alternateRowOffset, 0)
        ry = ry + np.where(_odd(x), cellShape.alternateColOffset, 0)
    else:
        rx += cellShape.alternateRowOffset if _odd(y) else 0
        ry += cellShape.alternateColOffset if _odd(x) else 0
    size = cellShape.getSize()
    return Rect(rx, ry, size, size)


def getFlipStateAt(cellShape, x, y):
    """GridUtility::getFlipStateAt (GridUtility.cpp:118-132)."""
    ox, oy = _odd(x), _odd(y)
    h = (cellShape.alternateColFlipHorizontal & ox) ^ (cellShape.alternateRowFlipHorizontal & oy)
    v = (cellShape.alternateColFlipVertical & ox) ^ (cellShape.alternateRowFlipVertical & oy)
    return FlipState(h, v)


def getGridState(cellGroup, imageRows, imageCols):
    """GridGenerator::getGridState for size steps 0: ``[StepBestFit]`` with 0 where
    the cell rect meets the image and NULLOPT where it does not
    (GridGenerator.cpp:61-70 over :113-140)."""
    shape = cellGroup.getCell(0)
    if shape.empty():
        return []
    gw, gh = calculateGridSize(shape, imageCols, imageRows)
    gx, gy = np.meshgrid(np.arange(gw, dtype=np.int64) - PAD_GRID,
                         np.arange(gh, dtype=np.int64) - PAD_GRID)
    r = getRectAt(shape, gx, gy)
    y0, y1 = np.clip(r.y, 0, imageRows), np.clip(r.y + r.height, 0, imageRows)
    x0, x1 = np.clip(r.x, 0, imageCols), np.clip(r.x + r.width, 0, imageCols)
    valid = (y0 != y1) & (x0 != x1)
    return [np.where(valid, 0, NULLOPT).astype(np.int64)]


def cell_records(cellGroup, stepBestFit, imageRows, imageCols):
    """One ``mm_cell`` per valid cell, in the reference's raster order
    (CPUPhotomosaicGenerator.cpp:63-77).

    Per cell: the global rect origin (getRectAt), the local bound at detail 1
    with the truncation / min / max guards of getCellAt
    (PhotomosaicGeneratorBase.cpp:299-326), the mask variant ``flipH + 2*flipV``
    and the padded grid position."""
    shape = cellGroup.getCell(0)
    detail = cellGroup.getDetail()
    size = cellGroup.getCell(0, True).getSize()
    gy, gx = np.nonzero(np.asarray(stepBestFit) >= 0)      # row-major == raster order
    x = gx.astype(np.int64) - PAD_GRID
    y = gy.astype(np.int64) - PAD_GRID
    r = getRectAt(shape, x, y)
    y0, y1 = np.clip(r.y, 0, imageRows), np.clip(r.y + r.height, 0, imageRows)
    x0, x1 = np.clip(r.x, 0, imageCols), np.clip(r.x + r.width, 0, imageCols)
    lx, ly, lw, lh = x0 - r.x, y0 - r.y, x1 - x0, y1 - y0
    bx = np.minimum(size - 1, (lx * detail).astype(np.int64))
    by = np.minimum(size - 1, (ly * detail).astype(np.int64))
    bw = np.maximum(1, (lw * detail).astype(np.int64))
    bh = np.maximum(1, (lh * detail).astype(np.int64))
    bw = np.minimum(size - bx, bw)
    bh = np.minimum(size - by, bh)
    flip = getFlipStateAt(shape, x, y)

    rec = np.zeros(len(x), dtype=CELL_DTYPE)
    rec["x"], rec["y"] = r.x, r.y
    rec["row_min"], rec["row_max"] = by, by + bh
    rec["col_min"], rec["col_max"] = bx, bx + bw
    rec["mask_id"] = flip.horizontal.astype(np.int64) + 2 * flip.vertical.astype(np.int64)
    rec["grid_x"], rec["grid_y"] = gx, gy
    return rec
