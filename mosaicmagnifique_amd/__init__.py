"""MI355X-native photomosaic best-fit engine behind the MosaicMagnifique generator interface.

The product is ``libmosaic_hip.so`` (hand-written HIP for gfx950 behind the C ABI
of ``include/mosaic_hip.h``); this package is the host-side mirror of the
reference's generator / grid / cell-shape interface for that one path.
Importing the package does not load the library; the first engine does, and it
raises if the library is not built (there is no CPU fallback).
"""
from . import colourdifference as ColourDifference  # noqa: F401
from . import colourscheme as ColourScheme  # noqa: F401
from . import grid as GridUtility  # noqa: F401
from .cellshape import CellGroup, CellShape  # noqa: F401

__all__ = ["ColourDifference", "ColourScheme", "GridUtility", "CellShape", "CellGroup",
           "HIPPhotomosaicGenerator", "Engine"]


def __getattr__(name):
    if name == "HIPPhotomosaicGenerator":
        from .generator import HIPPhotomosaicGenerator
        return HIPPhotomosaicGenerator
    if name in ("Engine", "EngineError", "device_count"):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)
