#!/usr/bin/env python3
"""Facts about every cell shape the reference ships (Cells/*.mcs: data files, big-endian QDataStream
with a PNG-encoded 512 x 512 mask, src/CellShape/CellShape.cpp:321-434) as read by the package's own
reader: name, file version, mask size, active pixels, a digest of the thresholded mask, tiling
parameters, and what CellShape::resized makes of it at the bench's 64 px cell size.
tests/test_image_library.py holds the Python and the C++ reader to these facts.

Runs ONLY in the build container (reads /root/reference); the JSON is committed (facts, not files).
"""
import glob
import hashlib
import json
import os
import struct
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mosaicmagnifique_amd.cellshape import CellShape  # noqa: E402

SRC = "/root/reference/Cells"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cells_mcs.json")


def facts(path):
    shape = CellShape.loadFromFile(path)
    mask = shape.getCellMask()
    small = shape.resized(64)
    raw = open(path, "rb").read()
    return {
        "file": os.path.basename(path),
        "file_bytes": len(raw),
        "file_sha256": hashlib.sha256(raw).hexdigest(),
        "version": struct.unpack_from(">I", raw, 4)[0],
        "name": shape.name,
        "size": shape.getSize(),
        "active": int((mask > 127).sum()),
        "mask_sha256": hashlib.sha256((mask > 127).astype("u1").tobytes()).hexdigest(),
        "params": {k: (bool(v) if isinstance(v, bool) else v) for k, v in shape.to_dict().items() if k != "name"},
        "resized64": {"active": int((small.getCellMask() != 0).sum()),
                      "params": {k: v for k, v in small.to_dict().items() if k != "name"}},
    }


def main():
    doc = {"source": "Cells/*.mcs of the reference checkout, read by mosaicmagnifique_amd.cellshape.CellShape.loadFromFile",
           "shapes": [facts(p) for p in sorted(glob.glob(os.path.join(SRC, "*.mcs")))]}
    json.dump(doc, open(OUT, "w"), indent=1)
    for s in doc["shapes"]:
        print(s["file"], s["name"], s["size"], s["active"], s["params"])


if __name__ == "__main__":
    main()
