#!/usr/bin/env python3
"""Decodes the reference's shipped cell shape Cells/Hexagon.mcs (data file, 512x512
mask + tiling parameters) and writes the 64x64 detail cell BASELINE config C uses
(CellShape::resized, src/CellShape/CellShape.cpp:281-312) as a small fixture.

Runs ONLY in the build container (reads /root/reference); the JSON is committed.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mosaicmagnifique_amd.cellshape import CellShape  # noqa: E402

SRC = "/root/reference/Cells/Hexagon.mcs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hexagon64.json")


def main():
    full = CellShape.loadFromFile(SRC)
    small = full.resized(64)
    mask = small.getCellMask(False, False)
    doc = {
        "source": "Cells/Hexagon.mcs (reference v3.4.2), resized to 64 by box mean + threshold > 127",
        "full": dict(full.to_dict(), size=full.getSize(), active=int((full.getCellMask() != 0).sum())),
        "params": small.to_dict(),
        "size": small.getSize(),
        "active": int((mask != 0).sum()),
        "rows": ["".join("1" if v else "0" for v in row) for row in mask],
    }
    json.dump(doc, open(OUT, "w"), indent=1)
    print({k: doc[k] for k in ("full", "params", "size", "active")})


if __name__ == "__main__":
    main()
