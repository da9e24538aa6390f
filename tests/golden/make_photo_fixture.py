#!/usr/bin/env python3
"""Photographic content for bench.py --config P (derived DATA, not source; run in the build container, CPU only).

The reference ships five photographs (SampleImages/*.jpg).  The synthetic benchmark content (synth.py: random colour
patches +- 32 of noise against a smooth target) lets the filter's early-exit bound drop half of the candidates before a
pixel is read; how much of the headline that is worth depends on the content, so one configuration runs on photographs:

  photo_target.png   annie-spratt-88306 box-resized to 2048 x 1363 (32 x 22 cells of 64 px, the last row clipped to 19 px)
  photo_library.png  1024 tiles of 64 x 64 cut on a regular lattice from the OTHER four photographs, each first box-resized
                     by 1/3 (a tile then shows about what a thumbnail-sized library image shows of a photograph), as a
                     32 x 32 sprite sheet; tile i is sheet[(i // 32) * 64 : .., (i % 32) * 64 : ..]
  photo_fixture.json sizes, source names, sha256 of the decoded u8 arrays (BGR order, as the engine takes them)

    python tests/golden/make_photo_fixture.py /root/reference/SampleImages
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
TARGET = "annie-spratt-88306-unsplash.jpg"
OTHERS = ["anthony-tran-600575-unsplash.jpg", "edgar-perez-424673-unsplash.jpg",
          "linnea-sandbakk-91258-unsplash.jpg", "roi-dimor-310628-unsplash.jpg"]
TILE, N_TILES, SHEET = 64, 1024, 32


def load():
    """(target u8 BGR [1363][2048][3], library u8 BGR [1024][64][64][3]) from the committed PNGs."""
    t = np.asarray(Image.open(os.path.join(HERE, "photo_target.png")).convert("RGB"))[:, :, ::-1]
    sheet = np.asarray(Image.open(os.path.join(HERE, "photo_library.png")).convert("RGB"))[:, :, ::-1]
    lib = sheet.reshape(SHEET, TILE, SHEET, TILE, 3).transpose(0, 2, 1, 3, 4).reshape(N_TILES, TILE, TILE, 3)
    return np.ascontiguousarray(t), np.ascontiguousarray(lib)


def main(src):
    im = Image.open(os.path.join(src, TARGET)).convert("RGB")
    w = 2048
    h = round(im.size[1] * w / im.size[0])
    target = im.resize((w, h), Image.BOX)
    tiles = []
    for name in OTHERS:
        ph = Image.open(os.path.join(src, name)).convert("RGB")
        ph = np.asarray(ph.resize((ph.size[0] // 3, ph.size[1] // 3), Image.BOX))
        for y in range(0, ph.shape[0] - TILE + 1, TILE):
            for x in range(0, ph.shape[1] - TILE + 1, TILE):
                tiles.append(ph[y:y + TILE, x:x + TILE])
    # a regular subsample of the lattice tiles, every photograph represented in proportion to its area
    pick = np.linspace(0, len(tiles) - 1, N_TILES).round().astype(int)
    assert len(np.unique(pick)) == N_TILES, len(tiles)
    lib = np.stack([tiles[i] for i in pick])
    sheet = lib.reshape(SHEET, SHEET, TILE, TILE, 3).transpose(0, 2, 1, 3, 4).reshape(SHEET * TILE, SHEET * TILE, 3)
    target.save(os.path.join(HERE, "photo_target.png"), optimize=True)
    Image.fromarray(sheet).save(os.path.join(HERE, "photo_library.png"), optimize=True)
    t_bgr, lib_bgr = load()
    assert np.array_equal(lib_bgr[:, :, :, ::-1], lib)
    meta = dict(target=TARGET, target_size=[h, w], library_from=OTHERS, tiles_available=len(tiles), tiles=N_TILES, tile=TILE,
                resize="PIL Image.BOX (area mean); library photographs by 1/3, target to width 2048",
                target_sha256=hashlib.sha256(t_bgr.tobytes()).hexdigest(), library_sha256=hashlib.sha256(lib_bgr.tobytes()).hexdigest())
    json.dump(meta, open(os.path.join(HERE, "photo_fixture.json"), "w"), indent=1)
    print(meta)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/SampleImages")
