"""Host-side image helpers the path's callers use (src/Other/ImageUtility.{h,cpp}).

The reference does these with OpenCV, which is not part of its checkout; the
functions below restate the library's published algorithms (OpenCV 4.x imgproc:
resize.cpp ``ResizeAreaFastVec`` / ``ResizeAreaFast_`` / ``computeResizeAreaTab``
/ ``ResizeArea_Invoker``; color ``RGB2Gray<uchar>``) for the cases the path
needs.  No reference test pins values at these steps: engine
(csrc/kernels.hpp), this module and the CPU checker used by the tests share ONE
specification, and the parity contract is stated relative to it.
"""
import math

import numpy as np

MAX_ENTROPY = 8.0          # ImageUtility.h:69


def shrink_u8(img, f):
    """INTER_AREA by a whole factor on u8: f = 2 -> (sum + 2) >> 2, else
    cvRound(float(sum) * (1.f / (f f))) (round half to even)."""
    img = np.asarray(img, dtype=np.uint8)
    if f == 1:
        return img.copy()
    r, c = img.shape[0] // f, img.shape[1] // f
    blocks = img[:r * f, :c * f].reshape((r, f, c, f) + img.shape[2:]).astype(np.int32)
    s = blocks.sum(axis=(1, 3))
    if f == 2:
        out = (s + 2) >> 2
    else:
        out = np.rint(s.astype(np.float32) * np.float32(1.0 / np.float32(f * f)))
    return np.clip(out, 0, 255).astype(np.uint8)


def shrink_f32(img, f):
    """INTER_AREA by a whole factor on f32 (last axis = channels): float row sums left to
    right, rows top to bottom, times 1/(f f) -- the arithmetic of cell_pixel (kernels.hpp)."""
    img = np.asarray(img, dtype=np.float32)
    if f == 1:
        return img.copy()
    r, c = img.shape[0] // f, img.shape[1] // f
    total = np.zeros((r, c) + img.shape[2:], dtype=np.float32)
    for i in range(f):
        row = np.zeros_like(total)
        for j in range(f):
            row = row + img[i:r * f:f, j:c * f:f]
        total = total + row
    return total * np.float32(1.0 / np.float32(f * f))


def _area_table(ssize, dsize, scale):
    """computeResizeAreaTab: (dst index, src index, weight f32) triples."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area_u8(img, rows, cols):
    """cv::resize(img, Size(cols, rows), INTER_AREA) for a shrink of a u8 image."""
    img = np.asarray(img, dtype=np.uint8)
    sr, sc = img.shape[:2]
    if (sr, sc) == (rows, cols):
        return img.copy()
    if rows > sr or cols > sc:
        raise NotImplementedError("INTER_CUBIC enlargement is outside the hot path")
    if sr % rows == 0 and sc % cols == 0 and sr // rows == sc // cols:
        return shrink_u8(img, sr // rows)
    xt = _area_table(sc, cols, 1.0 / (cols / sc))
    yt = _area_table(sr, rows, 1.0 / (rows / sr))
    src = img.astype(np.float32)
    out = np.empty((rows, cols) + img.shape[2:], dtype=np.uint8)
    acc = np.zeros((cols,) + img.shape[2:], dtype=np.float32)
    prev = yt[0][0]
    for dy, sy, beta in yt:
        buf = np.zeros_like(acc)
        for dx, sx, alpha in xt:                     # horizontal pass, in table order
            buf[dx] = buf[dx] + src[sy, sx] * alpha
        if dy != prev:
            out[prev] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
            acc = beta * buf
            prev = dy
        else:
            acc = acc + beta * buf
    out[prev] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return out


def bgr_to_gray_u8(img):
    """cv::cvtColor(BGR2GRAY) on u8: (B 1868 + G 9617 + R 4899 + 2^13) >> 14."""
    a = np.asarray(img, dtype=np.uint8).astype(np.int32)
    return ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def calculateEntropy(img_bgr, mask=None):
    """ImageUtility::calculateEntropy (ImageUtility.cpp:189-242)."""
    img_bgr = np.asarray(img_bgr)
    if img_bgr.size == 0:
        return 0.0
    gray = bgr_to_gray_u8(img_bgr)
    if mask is not None:
        mask = np.asarray(mask)
        if mask.shape != gray.shape:
            return 0.0
        gray = gray[mask != 0]
    hist = np.bincount(gray.ravel(), minlength=256)
    count = int(hist.sum())
    entropy = 0.0
    for value in hist:
        p = value / float(count) if count else 0.0
        if p > 0:
            entropy -= p * math.log2(p)
    return entropy
