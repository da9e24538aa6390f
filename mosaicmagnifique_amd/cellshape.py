"""Cell shapes: the binary mask + tiling parameters the best-fit path reads.

Host-side mirror of the reference's ``CellShape`` / ``CellGroup``
(src/CellShape/CellShape.{h,cpp}, src/CellShape/CellGroup.{h,cpp}), limited to
what feeds the hot path: the 4 flip variants of the mask, size, spacings,
offsets, alternate-flip flags, ``resized`` and the ``.mcs`` reader.  Editing
widgets, edge cells and the writer are out of scope.
"""
import io
import math
import struct

import numpy as np

from .imageutil import resize_area_u8

MCS_MAGIC = 0x87AECFB1          # CellShape.h; header check CellShape.cpp:392-395
MCS_VERSION = 8
MCS_VERSION_ENCODED = 8         # images PNG-encoded from v8, raw before (CellShape.cpp:411-414)


def _threshold(mask):
    """cv::threshold(mask, 127, 255, THRESH_BINARY), CellShape.cpp:118-123."""
    return np.where(np.asarray(mask) > 127, 255, 0).astype(np.uint8)


def _resize_area_u8(img, size):
    """INTER_AREA shrink of a square u8 image (what resizeImage picks for a factor < 1,
    Other/ImageUtility.cpp:50) -- imageutil.resize_area_u8, the restated OpenCV algorithm;
    parity at this step is UNPINNED."""
    if size > img.shape[0]:
        raise NotImplementedError("INTER_CUBIC enlargement is outside the hot path")
    return resize_area_u8(img, size, size)


class CellShape:
    """Binary mask + tiling geometry (CellShape.h:25-113)."""

    def __init__(self, mask=None):
        self.name = ""
        self._mask = None
        self.rowSpacing = self.colSpacing = 0
        self.alternateRowSpacing = self.alternateColSpacing = 0
        self.alternateRowOffset = self.alternateColOffset = 0
        self.alternateColFlipHorizontal = self.alternateColFlipVertical = False
        self.alternateRowFlipHorizontal = self.alternateRowFlipVertical = False
        if mask is not None:
            if np.isscalar(mask):                      # CellShape(size_t): all-255 square
                mask = np.full((int(mask), int(mask)), 255, dtype=np.uint8)
            mask = np.asarray(mask)
            # CellShape(const cv::Mat&): spacings default to the mask size (CellShape.cpp:36-44)
            self.rowSpacing = self.alternateRowSpacing = mask.shape[0]
            self.colSpacing = self.alternateColSpacing = mask.shape[1]
            self.setCellMask(mask)

    # -- mask ---------------------------------------------------------------
    def setCellMask(self, mask):
        """Threshold > 127 and pad to square (CellShape.cpp:113-135)."""
        mask = np.asarray(mask)
        if mask.dtype != np.uint8 or mask.ndim != 2:
            raise ValueError("Unsupported mask type")
        m = _threshold(mask)
        r, c = m.shape
        if r != c:                                      # imageToSquare(PAD)
            n = max(r, c)
            sq = np.zeros((n, n), dtype=np.uint8)
            oy, ox = (n - r) // 2, (n - c) // 2
            sq[oy:oy + r, ox:ox + c] = m
            m = sq
        self._mask = np.ascontiguousarray(m)

    def getCellMask(self, flippedHorizontal=False, flippedVertical=False):
        """The pre-flipped variant (CellShape.cpp:138-152)."""
        m = self._mask
        if flippedHorizontal:
            m = m[:, ::-1]
        if flippedVertical:
            m = m[::-1, :]
        return np.ascontiguousarray(m)

    def masks4(self):
        """The 4 masks in the order the device path indexes them,
        ``flipH + 2*flipV`` (CUDAPhotomosaicGenerator.cpp:122-134, :208)."""
        return np.stack([self.getCellMask(0, 0), self.getCellMask(1, 0),
                         self.getCellMask(0, 1), self.getCellMask(1, 1)])

    def getSize(self):
        return 0 if self._mask is None else int(self._mask.shape[0])

    def empty(self):
        return self._mask is None or self._mask.size == 0

    # -- geometry -----------------------------------------------------------
    def resized(self, size):
        """CellShape::resized (CellShape.cpp:281-312): resize + re-threshold the
        mask, floor the spacings (never < 1) and offsets, copy the flip flags."""
        size = int(size)
        if self.empty() or size == self.getSize():
            return self.copy()
        out = CellShape(_resize_area_u8(self._mask, size))
        ratio = float(size) / self._mask.shape[0]
        out.name = self.name
        out.rowSpacing = max(int(math.floor(self.rowSpacing * ratio)), 1)
        out.colSpacing = max(int(math.floor(self.colSpacing * ratio)), 1)
        out.alternateRowSpacing = max(int(math.floor(self.alternateRowSpacing * ratio)), 1)
        out.alternateColSpacing = max(int(math.floor(self.alternateColSpacing * ratio)), 1)
        out.alternateRowOffset = int(math.floor(self.alternateRowOffset * ratio))
        out.alternateColOffset = int(math.floor(self.alternateColOffset * ratio))
        out.alternateColFlipHorizontal = self.alternateColFlipHorizontal
        out.alternateColFlipVertical = self.alternateColFlipVertical
        out.alternateRowFlipHorizontal = self.alternateRowFlipHorizontal
        out.alternateRowFlipVertical = self.alternateRowFlipVertical
        return out

    def copy(self):
        out = CellShape()
        out.__dict__.update(self.__dict__)
        if self._mask is not None:
            out._mask = self._mask.copy()
        return out

    # -- .mcs ---------------------------------------------------------------
    @classmethod
    def loadFromFile(cls, filename):
        """Reads a ``.mcs`` (CellShape.cpp:373-434): big-endian QDataStream --
        u32 magic, u32 version, QString name, mask (PNG QByteArray from v8, raw
        type/rows/cols + bytes for v7; CustomQDataStream.h:22-87), 6 x i32
        spacings/offsets, 4 x bool flips."""
        data = open(filename, "rb").read()
        pos = 0

        def take(fmt):
            nonlocal pos
            v = struct.unpack_from(">" + fmt, data, pos)
            pos += struct.calcsize(">" + fmt)
            return v if len(v) > 1 else v[0]

        if take("I") != MCS_MAGIC:
            raise ValueError("File is not a valid .mcs: %s" % filename)
        version = take("I")
        if version > MCS_VERSION:
            raise ValueError(".mcs uses a newer file version: %s" % filename)
        if version < 7:
            raise ValueError(".mcs uses an outdated file version: %s" % filename)

        n = take("I")                                   # QString: byte length, UTF-16BE
        name = ""
        if n != 0xFFFFFFFF:
            name = data[pos:pos + n].decode("utf-16-be")
            pos += n

        if version >= MCS_VERSION_ENCODED:
            n = take("I")
            blob = data[pos:pos + n]
            pos += n
            from PIL import Image
            mask = np.array(Image.open(io.BytesIO(blob)))
        else:
            _type, rows, cols = take("III")
            n = take("I")
            mask = np.frombuffer(data[pos:pos + n], dtype=np.uint8).reshape(rows, cols)
            pos += n
        if mask.ndim == 3:
            mask = mask[..., 0]

        shape = cls()
        shape.name = name
        shape._mask = np.ascontiguousarray(mask.astype(np.uint8))   # loaded as stored
        shape.rowSpacing, shape.colSpacing = take("ii")
        shape.alternateRowSpacing, shape.alternateColSpacing = take("ii")
        shape.alternateRowOffset, shape.alternateColOffset = take("ii")
        (shape.alternateColFlipHorizontal, shape.alternateColFlipVertical,
         shape.alternateRowFlipHorizontal, shape.alternateRowFlipVertical) = \
            [bool(b) for b in take("BBBB")]
        return shape

    def saveToFile(self, filename):
        """CellShape::saveToFile (CellShape.cpp:321-370), version 8 (PNG-encoded mask)."""
        from PIL import Image
        with open(filename, "wb") as out:
            out.write(struct.pack(">II", MCS_MAGIC, MCS_VERSION))
            raw = self.name.encode("utf-16-be")
            out.write(struct.pack(">I", len(raw)))
            out.write(raw)
            png = io.BytesIO()
            Image.fromarray(self._mask, mode="L").save(png, format="PNG")
            blob = png.getvalue()
            out.write(struct.pack(">I", len(blob)))
            out.write(blob)
            out.write(struct.pack(">iiiiii", self.rowSpacing, self.colSpacing, self.alternateRowSpacing,
                                  self.alternateColSpacing, self.alternateRowOffset, self.alternateColOffset))
            out.write(struct.pack(">BBBB", self.alternateColFlipHorizontal, self.alternateColFlipVertical,
                                  self.alternateRowFlipHorizontal, self.alternateRowFlipVertical))

    def to_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "_mask"}

    @classmethod
    def from_dict(cls, params, mask):
        shape = cls()
        shape.__dict__.update(params)
        shape._mask = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        return shape


class CellGroup:
    """Normal + detail cell shapes per size step (CellGroup.h:26-64, CellGroup.cpp).

    The engine covers detail levels and size steps whose cell sizes divide evenly (detail
    100/n %, cell size a multiple of 2^steps): the resizes are then whole-factor box means.
    Other ratios need OpenCV's fractional INTER_AREA / INTER_CUBIC of cell pixels and stay
    on the reference host."""

    def __init__(self, cellShape=None):
        self.detail = 1.0
        self.sizeSteps = 0
        self._cells = [cellShape if cellShape is not None else CellShape()]
        self._detailCells = [self._cells[0]]
        if cellShape is not None:
            self.setDetail(100, reset=True)

    def setCellShape(self, cellShape):
        """CellGroup::setCellShape (CellGroup.cpp:29-56; edge cells are GUI-only)."""
        self._cells[0] = cellShape
        if not cellShape.empty():
            self.setDetail(0, reset=True)
        else:
            self._cells, self._detailCells = [cellShape], [cellShape]

    def setDetail(self, detail_percent=100, reset=False):
        """CellGroup::setDetail (CellGroup.cpp:65-87)."""
        new = self.detail if reset else detail_percent / 100.0
        if new != self.detail or reset:
            self.detail = new
            if not self._cells[0].empty():
                size = max(int(self._cells[0].getSize() * self.detail), 1)
                self._detailCells[0] = self._cells[0].resized(size)
            self.setSizeSteps(self.sizeSteps, reset=True)

    def getDetail(self):
        return self.detail

    def setSizeSteps(self, steps, reset=False):
        """CellGroup::setSizeSteps (CellGroup.cpp:96-128): each step halves the cell."""
        steps = int(steps)
        self._cells = (self._cells + [None] * steps)[:steps + 1]
        self._detailCells = (self._detailCells + [None] * steps)[:steps + 1]
        size = self._cells[0].getSize()
        for step in range(1, steps + 1):
            size //= 2
            if reset or step > self.sizeSteps or self._cells[step] is None:
                self._cells[step] = self._cells[step - 1].resized(size)
                self._detailCells[step] = self._cells[step].resized(max(int(size * self.detail), 1))
        self.sizeSteps = steps

    def getSizeSteps(self):
        return self.sizeSteps

    def getCell(self, sizeStep=0, detail=False):
        return (self._detailCells if detail else self._cells)[sizeStep]

    def getCellSize(self, sizeStep=0, detail=False):
        return self.getCell(sizeStep, detail).getSize()

    def cellScale(self, sizeStep=0):
        """Normal-cell pixels per detail pixel and axis at a step when that is a whole number,
        else 0 (the engine then uses the general INTER_AREA tables, mm_set_cell_resize)."""
        n, d = self.getCellSize(sizeStep), self.getCellSize(sizeStep, True)
        return n // d if d > 0 and n % d == 0 else 0
