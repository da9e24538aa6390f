"""Colour-difference selector of the best-fit path.

Mirrors ``ColourDifference::Type`` and ``Type_STR`` of the reference
(src/Photomosaic/ColourDifference.h:13-25): the integer values are what crosses
the C ABI as ``diff_type``.
"""
import enum


class Type(enum.IntEnum):
    RGB_EUCLIDEAN = 0
    CIE76 = 1
    CIEDE2000 = 2
    MAX = 3


Type_STR = ("RGB Euclidean", "CIE76", "CIEDE2000")


def strToEnum(name):
    """ColourDifference::strToEnum (ColourDifference.cpp:4-13): MAX when unknown."""
    for i, s in enumerate(Type_STR):
        if s == name:
            return Type(i)
    return Type.MAX


def uses_lab(t):
    """CIE76 and CIEDE2000 work on CIELab triplets
    (PhotomosaicGeneratorBase.cpp:238-244, :276-282)."""
    return Type(t) in (Type.CIE76, Type.CIEDE2000)
