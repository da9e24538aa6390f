"""Colour-scheme variants of the main image.

Mirrors ``ColourScheme`` of the reference (src/Photomosaic/ColourScheme.h,
ColourScheme.cpp:7-177): ``Type`` / ``Type_STR`` / ``strToEnum`` / ``getFunction`` and
the six ``getColourScheme*`` functions, each returning the original image followed by
its hue-rotated variants.

A variant is: u8 -> f32 (values stay 0..255) -> ``cvtColor(BGR2HSV_FULL)`` ->
``H = fmod(H + degrees, 360)`` -> ``cvtColor(HSV2BGR_FULL)`` -> ``convertTo(u8)``.
OpenCV (4.5.2) is not part of the reference checkout and no reference test holds
vectors for these conversions, so the arithmetic is *specified* (csrc/kernels.hpp
``hue_rotate_px``, after OpenCV's documented f32 formulas): parity unpinned at this
step, as for BGR2Lab.  ``hue_rotate_u8`` below is the numpy restatement in the same f32
operation order; the engine (``Engine.hue_rotate_u8``) and the oracle agree with it bit
for bit (tests/test_gpu_colour_scheme.py).

How the generators use the variants (and why only the LAST one matters) is in
``HIPPhotomosaicGenerator.setColourScheme``.
"""
import enum

import numpy as np


class Type(enum.IntEnum):
    NONE = 0
    COMPLEMENTARY = 1
    TRIADIC = 2
    COMPOUND = 3
    TETRADIC = 4
    ANALAGOUS = 5
    MAX = 6


Type_STR = ("None", "Complementary", "Triadic", "Compound", "Tetradic", "Analagous")

# hue rotations of the variants after the original image (ColourScheme.cpp:44-177)
ROTATIONS = {
    Type.NONE: (),
    Type.COMPLEMENTARY: (180.0,),
    Type.TRIADIC: (120.0, 240.0),
    Type.COMPOUND: (150.0, 210.0),
    Type.TETRADIC: (90.0, 180.0, 270.0),
    Type.ANALAGOUS: (30.0, 60.0, 90.0),
}


def strToEnum(name):
    """ColourScheme::strToEnum (ColourScheme.cpp:7-16): MAX when unknown."""
    for i, s in enumerate(Type_STR):
        if s == name:
            return Type(i)
    return Type.MAX


_F = np.float32
_EPS = np.float32(1.1920928955078125e-07)          # FLT_EPSILON
_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def hue_rotate_u8(image, degrees):
    """The variant of ``image`` (u8 BGR) with its hue rotated by ``degrees``; f32 throughout,
    the operation order of ``hue_rotate_px`` (csrc/kernels.hpp)."""
    img = np.asarray(image, dtype=np.uint8)
    flat = img.reshape(-1, 3).astype(_F)
    b, g, r = flat[:, 0], flat[:, 1], flat[:, 2]
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    s = diff / (np.abs(v) + _EPS)
    k = _F(60.0) / (diff + _EPS)
    h = np.where(v == r, (g - b) * k, np.where(v == g, (b - r) * k + _F(120.0), (r - g) * k + _F(240.0))).astype(_F)
    h = np.where(h < 0, h + _F(360.0), h).astype(_F)
    h = np.fmod(h + _F(degrees), _F(360.0)).astype(_F)
    hh = np.fmod(h * (_F(6.0) / _F(360.0)), _F(6.0)).astype(_F)
    sector = np.floor(hh).astype(np.int64)
    hh = (hh - sector.astype(_F)).astype(_F)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hh = np.where(bad, _F(0), hh).astype(_F)
    one = _F(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * hh), v * (one - s * (one - hh))], axis=1).astype(_F)
    out = np.take_along_axis(tab, _SECTOR[sector], axis=1)
    out = np.where((s == 0)[:, None], v[:, None], out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8).reshape(img.shape)      # rint: half to even, as cvRound


def getFunction(t):
    """ColourScheme::getFunction (ColourScheme.cpp:19-32): image -> [original, variants...]."""
    t = Type(t)
    if t == Type.MAX:
        raise ValueError("No function for given type")
    return lambda image, rotate=hue_rotate_u8: [np.asarray(image)] + [rotate(image, d) for d in ROTATIONS[t]]
