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

MCS_MAGIC = 0x87AECFB1          # CellShape.h; header check CellShape.cpp:392-395
MCS_VERSION = 8
MCS_VERSION_ENCODED = 8         # images PNG-encoded from v8, raw before (CellShape.cpp:411-414)


def _threshold(mask):
    """cv::threshold(mask, 127, 255, THRESH_BINARY), CellShape.cpp:118-123."""
    return np.where(np.asarray(mask) > 127, 255, 0).astype(np.uint8)


def _resize_area_u8(img, size):
    """INTER_AREA shrink of a square u8 image (what resizeImage picks for a
    factor < 1, Other/ImageUtility.cpp:50).  Area-weighted mean, rounded half
    to even like cv::saturate_cast<uchar>.  For integer factors this is the
    exact box mean; OpenCV's own arithmetic is not in the reference checkout,
    so parity at this step is UNPINNED -- oracle and engine are
    always given the same resized mask."""
    n = img.shape[0]
    if size > n:
        raise NotImplementedError("INTER_CUBIC enlargement is outside the hot path")
    edges = np.linspace(0.0, n, size + 1)
    w = np.zeros((size, n), dtype=np.float64)
    for i in range(size):
        lo, hi = edges[i], edges[i + 1]
        j0, j1 = int(math.floor(lo)), int(math.ceil(hi))
        for j in range(j0, min(j1, n)):
            w[i, j] = max(0.0, min(hi, j + 1) - max(lo, j))
        w[i] /= (hi - lo)
    out = w @ img.astype(np.float64) @ w.T
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


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
    """Normal + detail cell shape for size step 0 (CellGroup.h:26-64).

    The engine covers detail 100 % and size steps 0 (all BASELINE configs);
    other detail levels need OpenCV's resize of the cell pixels, which stays on
    the reference host."""

    def __init__(self, cellShape=None):
        self.detail = 1.0
        self.sizeSteps = 0
        self._cell = cellShape if cellShape is not None else CellShape()

    def setCellShape(self, cellShape):
        self._cell = cellShape

    def setDetail(self, detail_percent=100):
        if int(detail_percent) != 100:
            raise NotImplementedError("detail != 100 % resizes cell pixels with OpenCV on the host")
        self.detail = 1.0

    def getDetail(self):
        return self.detail

    def setSizeSteps(self, steps):
        if int(steps) != 0:
            raise NotImplementedError("size steps > 0 are outside the hot-path scope (SURVEY 8f N3)")
        self.sizeSteps = 0

    def getSizeSteps(self):
        return self.sizeSteps

    def getCell(self, sizeStep=0, detail=False):
        if sizeStep != 0:
            raise IndexError(sizeStep)
        return self._cell

    def getCellSize(self, sizeStep=0, detail=False):
        return self.getCell(sizeStep, detail).getSize()
