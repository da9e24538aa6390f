"""Grid geometry of the best-fit path (host side).

Mirror of ``GridUtility`` (src/Grid/GridUtility.{h,cpp}), ``GridBounds::mergeBounds``
and ``GridGenerator::getGridState`` / ``findCellState`` with the entropy subdivision
over size steps (src/Grid/GridGenerator.cpp:29-193), plus the marshalling of valid
cells into the ``mm_cell`` records the C ABI takes.

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


def mergeBounds(bounds):
    """GridBounds::mergeBounds (Grid/GridBounds.cpp:36-107) on a list of Rect; returns the merged list."""
    b = [Rect(*r) for r in bounds]

    def union(p, q):
        x0, y0 = min(p.x, q.x), min(p.y, q.y)
        x1, y1 = max(p.x + p.width, q.x + q.width), max(p.y + p.height, q.y + q.height)
        return Rect(x0, y0, x1 - x0, y1 - y0)

    merged = True
    while merged:
        merged = False
        i = 0
        while i + 1 < len(b):
            j = i + 1
            while j < len(b) and i + 1 < len(b):
                p, q = b[i], b[j]
                merge = False
                if p.x == q.x and p.width == q.width:
                    d = q.y - p.y
                    merge = d == 0 or (0 < d <= p.height) or (d < 0 and -d <= q.height)
                elif p.y == q.y and p.height == q.height:
                    d = q.x - p.x
                    merge = d == 0 or (0 < d <= p.width) or (d < 0 and -d <= q.width)
                else:
                    u = union(p, q)
                    merge = u == p or u == q
                if merge:
                    b[i] = union(p, q)
                    del b[j]
                    merged = True
                else:
                    j += 1
            if i + 1 < len(b):
                i += 1
            else:
                break
    return b


def _detail_bound(lx, ly, lw, lh, detail, size):
    """The local bound at the detail size: int truncation + the min / max guards of
    PhotomosaicGeneratorBase.cpp:318-326 (= GridGenerator.cpp:156-170).  Scalars or arrays."""
    trunc = (lambda v: v.astype(np.int64)) if isinstance(lx, np.ndarray) else int
    bx = np.minimum(size - 1, trunc(lx * detail))
    by = np.minimum(size - 1, trunc(ly * detail))
    bw = np.maximum(1, trunc(lw * detail))
    bh = np.maximum(1, trunc(lh * detail))
    return bx, by, np.minimum(size - bx, bw), np.minimum(size - by, bh)


def findCellState(cellGroup, mainImage, x, y, bounds, step):
    """GridGenerator::findCellState (GridGenerator.cpp:113-193) -> (state, entropy exceeded)."""
    from .imageutil import MAX_ENTROPY, calculateEntropy, resize_area_u8
    shape = cellGroup.getCell(step)
    g = getRectAt(shape, x, y)
    inside = False
    for b in bounds:
        y0, y1 = min(max(g.y, b.y), b.y + b.height), min(max(g.y + g.height, b.y), b.y + b.height)
        x0, x1 = min(max(g.x, b.x), b.x + b.width), min(max(g.x + g.width, b.x), b.x + b.width)
        if y0 != y1 and x0 != x1:
            inside = True
            break
    if not inside:
        return NULLOPT, False
    if mainImage is not None and step < cellGroup.getSizeSteps():
        rows, cols = mainImage.shape[0], mainImage.shape[1]
        y0, y1 = min(max(g.y, 0), rows), min(max(g.y + g.height, 0), rows)
        x0, x1 = min(max(g.x, 0), cols), min(max(g.x + g.width, 0), cols)
        size = cellGroup.getCellSize(step, True)
        bx, by, bw, bh = _detail_bound(x0 - g.x, y0 - g.y, x1 - x0, y1 - y0, cellGroup.getDetail(), size)
        flip = getFlipStateAt(shape, x, y)
        mask = cellGroup.getCell(step, True).getCellMask(flip.horizontal, flip.vertical)[by:by + bh, bx:bx + bw]
        cell = mainImage[y0:y1, x0:x1]
        if cell.shape[:2] != mask.shape:
            cell = resize_area_u8(cell, mask.shape[0], mask.shape[1])      # resizeImage(EXACT), :180-181
        if calculateEntropy(cell, mask) >= MAX_ENTROPY * 0.7:
            return NULLOPT, True
    return 0, False


def _meets_any_bound(bounds, qx, qy, qw, qh):
    """findCellState's in-bounds test (GridGenerator.cpp:124-140) for arrays of cell rects at once: does the rect
    share positive area with at least one bound?  The answer depends on the UNION of the bounds alone, and
    ``GridBounds::mergeBounds`` never changes that union (it joins rects whose union is again a rect), so no merging is
    needed here: the bounds are rasterised on the grid of their own edges (a difference array and two prefix sums) and
    every query is four look-ups.  ``bounds``: int array [n, 4] of (x, y, width, height)."""
    bounds = np.asarray(bounds, dtype=np.int64).reshape(-1, 4)
    if len(bounds) == 0:
        return np.zeros(np.shape(qx), dtype=bool)
    bx0, by0 = bounds[:, 0], bounds[:, 1]
    bx1, by1 = bx0 + bounds[:, 2], by0 + bounds[:, 3]
    xs, ys = np.unique(np.concatenate([bx0, bx1])), np.unique(np.concatenate([by0, by1]))
    nx, ny = len(xs) - 1, len(ys) - 1
    diff = np.zeros((ny + 1, nx + 1), dtype=np.int64)
    i0, i1, j0, j1 = np.searchsorted(xs, bx0), np.searchsorted(xs, bx1), np.searchsorted(ys, by0), np.searchsorted(ys, by1)
    np.add.at(diff, (j0, i0), 1)
    np.add.at(diff, (j0, i1), -1)
    np.add.at(diff, (j1, i0), -1)
    np.add.at(diff, (j1, i1), 1)
    covered = diff.cumsum(0).cumsum(1)[:ny, :nx] > 0
    pre = np.zeros((ny + 1, nx + 1), dtype=np.int64)
    pre[1:, 1:] = covered.cumsum(0).cumsum(1)
    a = np.clip(np.searchsorted(xs, qx, side="right") - 1, 0, nx)
    b = np.clip(np.searchsorted(xs, qx + qw, side="left"), 0, nx)
    c = np.clip(np.searchsorted(ys, qy, side="right") - 1, 0, ny)
    d = np.clip(np.searchsorted(ys, qy + qh, side="left"), 0, ny)
    b, d = np.maximum(a, b), np.maximum(c, d)
    return (pre[d, b] - pre[c, b] - pre[d, a] + pre[c, a]) > 0


def _grid_state_on_device(cellGroup, mainImage, engine):
    """getGridState's step loop with the per-cell image work (resize of the visible part, BGR2GRAY, masked
    histogram, entropy) on the device: ``engine.cell_entropy`` over all cells of a step that lie inside an active
    bound; the integer geometry and the bounds stay here, vectorised (``_meets_any_bound``)."""
    imageRows, imageCols = mainImage.shape[0], mainImage.shape[1]
    engine.set_main_image_bgr(mainImage)
    gridState = []
    active = np.array([[0, 0, imageCols, imageRows]], dtype=np.int64)
    steps = cellGroup.getSizeSteps()
    for step in range(steps + 1):
        if len(active) == 0:
            break
        shape = cellGroup.getCell(step)
        gw, gh = calculateGridSize(shape, imageCols, imageRows)
        gx, gy = np.meshgrid(np.arange(gw, dtype=np.int64) - PAD_GRID, np.arange(gh, dtype=np.int64) - PAD_GRID)
        r = getRectAt(shape, gx, gy)
        inside = _meets_any_bound(active, r.x, r.y, r.width + 0 * r.x, r.height + 0 * r.y)
        state = np.where(inside, 0, NULLOPT).astype(np.int64)
        active = np.zeros((0, 4), dtype=np.int64)
        if step < steps and inside.any():
            detail = cellGroup.getCell(step, True)
            engine.set_masks(detail.masks4())
            recs = cell_records(cellGroup, state, imageRows, imageCols, step)
            split = engine.cell_entropy(recs, shape.getSize()) != 0
            state[recs["grid_y"][split], recs["grid_x"][split]] = NULLOPT
            cx, cy = recs["x"][split].astype(np.int64), recs["y"][split].astype(np.int64)
            y0, y1 = np.clip(cy, 0, imageRows), np.clip(cy + shape.getSize(), 0, imageRows)      # :77-103
            x0, x1 = np.clip(cx, 0, imageCols), np.clip(cx + shape.getSize(), 0, imageCols)
            keep = (y0 != y1) & (x0 != x1)
            active = np.stack([x0, y0, x1 - x0, y1 - y0], axis=1)[keep]
        gridState.append(state)
    return gridState


def getGridState(cellGroup, imageRows, imageCols, mainImage=None, engine=None):
    """GridGenerator::getGridState (GridGenerator.cpp:29-110): one StepBestFit per size step;
    a cell whose masked grey-level entropy reaches 0.7 * MAX_ENTROPY is handed to the next
    (half-size) step.  ``mainImage``: u8 BGR, needed only with size steps > 0; without it
    (or with size steps 0) a cell is valid iff its rect meets the image.  ``engine``: an Engine
    whose ``cell_entropy`` does the per-cell image work on the device (it leaves the LAST step's
    detail masks staged on that engine)."""
    if engine is not None and mainImage is not None and cellGroup.getSizeSteps() > 0 and not cellGroup.getCell(0).empty():
        return _grid_state_on_device(cellGroup, np.asarray(mainImage), engine)
    if cellGroup.getCell(0).empty():
        return []
    if mainImage is not None:
        mainImage = np.asarray(mainImage)
        imageRows, imageCols = mainImage.shape[0], mainImage.shape[1]
    if cellGroup.getSizeSteps() == 0 or mainImage is None:
        # fast path: whole-image bound, no entropy test (:113-140,:191-192)
        shape = cellGroup.getCell(0)
        gw, gh = calculateGridSize(shape, imageCols, imageRows)
        gx, gy = np.meshgrid(np.arange(gw, dtype=np.int64) - PAD_GRID,
                             np.arange(gh, dtype=np.int64) - PAD_GRID)
        r = getRectAt(shape, gx, gy)
        y0, y1 = np.clip(r.y, 0, imageRows), np.clip(r.y + r.height, 0, imageRows)
        x0, x1 = np.clip(r.x, 0, imageCols), np.clip(r.x + r.width, 0, imageCols)
        valid = (y0 != y1) & (x0 != x1)
        return [np.where(valid, 0, NULLOPT).astype(np.int64)]

    gridState = []
    active = [Rect(0, 0, imageCols, imageRows)]
    for step in range(cellGroup.getSizeSteps() + 1):
        if not active:
            break
        shape = cellGroup.getCell(step)
        gw, gh = calculateGridSize(shape, imageCols, imageRows)
        state = np.full((gh, gw), NULLOPT, dtype=np.int64)
        nxt = []
        for y in range(-PAD_GRID, gh - PAD_GRID):
            for x in range(-PAD_GRID, gw - PAD_GRID):
                value, split = findCellState(cellGroup, mainImage, x, y, active, step)
                state[y + PAD_GRID, x + PAD_GRID] = value
                if split:
                    g = getRectAt(shape, x, y)
                    y0, y1 = min(max(g.y, 0), imageRows), min(max(g.y + g.height, 0), imageRows)
                    x0, x1 = min(max(g.x, 0), imageCols), min(max(g.x + g.width, 0), imageCols)
                    if y0 == y1 or x0 == x1:
                        continue
                    nxt.append(Rect(x0, y0, x1 - x0, y1 - y0))
        gridState.append(state)
        active = mergeBounds(nxt) if nxt else []
    return gridState


def cell_records(cellGroup, stepBestFit, imageRows, imageCols, step=0):
    """One ``mm_cell`` per valid cell of a size step, in the reference's raster order
    (CPUPhotomosaicGenerator.cpp:63-77).

    Per cell: the global rect origin of the NORMAL cell (getRectAt), the local bound at the
    detail size with the truncation / min / max guards of getCellAt
    (PhotomosaicGeneratorBase.cpp:299-326), the mask variant ``flipH + 2*flipV`` and the
    padded grid position."""
    shape = cellGroup.getCell(step)
    detail = cellGroup.getDetail()
    size = cellGroup.getCell(step, True).getSize()
    gy, gx = np.nonzero(np.asarray(stepBestFit) >= 0)      # row-major == raster order
    x = gx.astype(np.int64) - PAD_GRID
    y = gy.astype(np.int64) - PAD_GRID
    r = getRectAt(shape, x, y)
    y0, y1 = np.clip(r.y, 0, imageRows), np.clip(r.y + r.height, 0, imageRows)
    x0, x1 = np.clip(r.x, 0, imageCols), np.clip(r.x + r.width, 0, imageCols)
    bx, by, bw, bh = _detail_bound(x0 - r.x, y0 - r.y, x1 - x0, y1 - y0, detail, size)
    flip = getFlipStateAt(shape, x, y)

    rec = np.zeros(len(x), dtype=CELL_DTYPE)
    rec["x"], rec["y"] = r.x, r.y
    rec["row_min"], rec["row_max"] = by, by + bh
    rec["col_min"], rec["col_max"] = bx, bx + bw
    rec["mask_id"] = flip.horizontal.astype(np.int64) + 2 * flip.vertical.astype(np.int64)
    rec["grid_x"], rec["grid_y"] = gx, gy
    return rec
