"""Deterministic synthetic inputs for tests and bench.py.

Everything is a pure function of a counter-based 32-bit hash, so any process
(every rank of a multi-GPU run, the CPU oracle leg, the GPU leg) builds
byte-identical images without shipping data.

Hash: ``mix(seed, a, b, c, d)`` folds each 32-bit word in with the murmur3
finaliser (fmix32); uniform enough for image noise, trivially vectorised.
"""
import numpy as np

_U = np.uint32


def _fmix32(h):
    h = h ^ (h >> _U(16))
    h = h * _U(0x85EBCA6B)
    h = h ^ (h >> _U(13))
    h = h * _U(0xC2B2AE35)
    h = h ^ (h >> _U(16))
    return h


def mix(seed, *words):
    """hash32(seed, w0, w1, ...): each word broadcast against the others."""
    with np.errstate(over="ignore"):
        h = np.asarray(seed, dtype=np.uint32) * _U(0x9E3779B1) + _U(0x7F4A7C15)
        for w in words:
            w = np.asarray(w).astype(np.uint32)
            h = _fmix32(h ^ (w * _U(0x9E3779B1) + _U(0x165667B1)))
        return h


SEED_LIBRARY = 0x11B
SEED_TARGET = 0x7A6


def library_u8(n_images, size, seed=SEED_LIBRARY, first=0):
    """Library images ``first .. first+n_images-1`` as u8 BGR ``[n][size][size][3]``.

    Image i = a base colour (uniform in the BGR cube) + per-pixel noise in
    [-32, 32]: distinct, low-variance patches, so candidate scores spread
    realistically and accidental exact ties do not happen.
    """
    i = np.arange(first, first + n_images, dtype=np.uint32)
    c = np.arange(3, dtype=np.uint32)
    base = (mix(seed, i[:, None], c[None, :], 0xFFFF) % _U(256)).astype(np.int16)
    pix = np.arange(size * size, dtype=np.uint32)
    h = mix(seed, i[:, None], pix[None, :])                       # one hash per pixel ...
    out = np.empty((n_images, size * size, 3), dtype=np.uint8)
    with np.errstate(over="ignore"):
        for ch in range(3):                                       # ... three channels from it
            noise = (_fmix32(h + _U(0x27D4EB2F) * _U(ch + 1)) % _U(65)).astype(np.int16) - 32
            out[:, :, ch] = np.clip(base[:, ch:ch + 1] + noise, 0, 255)
    return out.reshape(n_images, size, size, 3)


def target_u8(rows, cols, lattice=64, seed=SEED_TARGET):
    """Target image u8 BGR ``[rows][cols][3]``: bilinear blend of a coarse
    random colour lattice (one node per ``lattice`` px) plus noise in [-16, 16]."""
    ly, lx = rows // lattice + 2, cols // lattice + 2
    gy = np.arange(ly, dtype=np.uint32)[:, None, None]
    gx = np.arange(lx, dtype=np.uint32)[None, :, None]
    c = np.arange(3, dtype=np.uint32)[None, None, :]
    nodes = (mix(seed, gy, gx, c, 0xA5A5) % _U(256)).astype(np.float64)

    yy = np.arange(rows, dtype=np.float64) / lattice
    xx = np.arange(cols, dtype=np.float64) / lattice
    y0 = np.floor(yy).astype(np.int64)
    x0 = np.floor(xx).astype(np.int64)
    fy = (yy - y0)[:, None, None]
    fx = (xx - x0)[None, :, None]
    fx2 = (xx - x0)[None, :, None]
    along_x = nodes[:, x0] * (1 - fx2) + nodes[:, x0 + 1] * fx2      # [ly][cols][3], separable blend
    del fx
    out = np.empty((rows, cols, 3), dtype=np.uint8)
    band = max(1, (1 << 22) // max(cols, 1))
    for r0 in range(0, rows, band):
        r1 = min(rows, r0 + band)
        f = fy[r0:r1]
        smooth = along_x[y0[r0:r1]] * (1 - f) + along_x[y0[r0:r1] + 1] * f
        y = np.arange(r0, r1, dtype=np.uint32)[:, None]
        x = np.arange(cols, dtype=np.uint32)[None, :]
        h = mix(seed, y, x, 0x5A5A)
        with np.errstate(over="ignore"):
            for ch in range(3):
                noise = (_fmix32(h + _U(0x27D4EB2F) * _U(ch + 1)) % _U(33)).astype(np.float64) - 16
                out[r0:r1, :, ch] = np.clip(np.rint(smooth[:, :, ch] + noise), 0, 255)
    return out


def bgr_u8_to_lab_f32(img):
    """Chunked driver of _lab_block (keeps the float64 temporaries cache-sized)."""
    img = np.asarray(img)
    flat = img.reshape(-1, 3)
    out = np.empty(flat.shape, dtype=np.float32)
    step = 1 << 20
    for i in range(0, flat.shape[0], step):
        out[i:i + step] = _lab_block(flat[i:i + step])
    return out.reshape(img.shape)


def _lab_block(img):
    """u8 BGR -> f32 CIELab (L in [0,100], a,b about [-127,127]); sRGB, D65.

    The reference does this with OpenCV (``convertTo(CV_32F, 1/255)`` then
    ``cvtColor(COLOR_BGR2Lab)``, PhotomosaicGeneratorBase.cpp:238-244); OpenCV's
    arithmetic is not part of the reference checkout and no reference test pins
    Lab values, so the parity contract starts AFTER this conversion: oracle and
    engine are fed the same f32 Lab arrays.  Plain float64 formulae, rounded
    to f32 once at the end.
    """
    lut = np.arange(256, dtype=np.float64) / 255.0
    lut = np.where(lut > 0.04045, ((lut + 0.055) / 1.055) ** 2.4, lut / 12.92)
    lin = lut[img]                                                # sRGB decode of u8 via table
    b, g, r = lin[..., 0], lin[..., 1], lin[..., 2]
    x = (0.412453 * r + 0.357580 * g + 0.180423 * b) / 0.950456
    y = 0.212671 * r + 0.715160 * g + 0.072169 * b
    z = (0.019334 * r + 0.119193 * g + 0.950227 * b) / 1.088754

    def f(t):
        return np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)

    fx, fy, fz = f(x), f(y), f(z)
    L = np.where(y > 0.008856, 116.0 * fy - 16.0, 903.3 * y)
    out = np.stack([L, 500.0 * (fx - fy), 200.0 * (fy - fz)], axis=-1)
    return out.astype(np.float32)


def preprocess(img_u8, diff_type):
    """preprocessMainImage / preprocessLibraryImages colour conversion
    (PhotomosaicGeneratorBase.cpp:223-252, :255-290): f32 BGR in [0,255] for
    RGB_EUCLIDEAN, f32 Lab for CIE76 / CIEDE2000."""
    from .colourdifference import uses_lab
    if uses_lab(diff_type):
        return bgr_u8_to_lab_f32(img_u8)
    return img_u8.astype(np.float32)
