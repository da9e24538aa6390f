#!/usr/bin/env python3
"""Photographic content at BASELINE config D's scale, for bench.py --config PD (derived DATA, not source; CPU only).

Config P (make_photo_fixture.py) holds 704 cells against 1024 tiles.  PD is config D's job shape -- 4096 x 4096 target, 64 x 64 cells
(4096 of them), 5000 library images, CIEDE2000, repeat range 4 (+500) -- on photographs:

  photo_pd_target.jpg    annie-spratt-88306: the centred square (2662 x 2662) of the photograph, Lanczos-resized to 4096 x 4096
  photo_pd_library.jpg   5000 tiles of 64 x 64 cut on regular lattices from the OTHER four photographs at TWO scales (box-resized by 1/2
                         and by 1/3: a library holds close-ups and wider views of the same scenes), a regular subsample of the 5310
                         lattice tiles, as a 100 x 50 sprite sheet; tile i is sheet[(i // 100) * 64 : .., (i % 100) * 64 : ..]
  photo_pd_fixture.json  sizes, source names, sha256 of the DECODED u8 arrays (BGR order, as the engine takes them)

JPEG (quality 92, 4:4:4) instead of PNG keeps the two files at a few MB; what the tests and the bench consume is the decoded array,
whose sha256 `load()` checks -- a decoder that rounds differently fails loudly instead of silently invalidating the golden
(tests/golden/grid_PD.npz is keyed on the sha256 of the f32 Lab arrays made from these).

    python tests/golden/make_photo_fixture_pd.py /root/reference/SampleImages
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
TILE, N_TILES, SHEET_W, SHEET_H, SIDE = 64, 5000, 100, 50, 4096


def _decode():
    t = np.asarray(Image.open(os.path.join(HERE, "photo_pd_target.jpg")).convert("RGB"))[:, :, ::-1]
    sheet = np.asarray(Image.open(os.path.join(HERE, "photo_pd_library.jpg")).convert("RGB"))[:, :, ::-1]
    lib = sheet.reshape(SHEET_H, TILE, SHEET_W, TILE, 3).transpose(0, 2, 1, 3, 4).reshape(N_TILES, TILE, TILE, 3)
    return np.ascontiguousarray(t), np.ascontiguousarray(lib)


def load(check=True):
    """(target u8 BGR [4096][4096][3], library u8 BGR [5000][64][64][3]) from the committed JPEGs, sha256-checked."""
    t, lib = _decode()
    if check:
        meta = json.load(open(os.path.join(HERE, "photo_pd_fixture.json")))
        got = (hashlib.sha256(t.tobytes()).hexdigest(), hashlib.sha256(lib.tobytes()).hexdigest())
        if got != (meta["target_sha256"], meta["library_sha256"]):
            raise RuntimeError("photo_pd_*.jpg decode to other bytes here than where the fixture was made (another libjpeg?): "
                               "the PD golden does not apply")
    return t, lib


def main(src):
    im = Image.open(os.path.join(src, TARGET)).convert("RGB")
    w, h = im.size
    s = min(w, h)
    box = ((w - s) // 2, (h - s) // 2, (w - s) // 2 + s, (h - s) // 2 + s)
    target = im.crop(box).resize((SIDE, SIDE), Image.LANCZOS)
    tiles = []
    for name in OTHERS:
        full = Image.open(os.path.join(src, name)).convert("RGB")
        for div in (2, 3):
            ph = np.asarray(full.resize((full.size[0] // div, full.size[1] // div), Image.BOX))
            for y in range(0, ph.shape[0] - TILE + 1, TILE):
                for x in range(0, ph.shape[1] - TILE + 1, TILE):
                    tiles.append(ph[y:y + TILE, x:x + TILE])
    pick = np.linspace(0, len(tiles) - 1, N_TILES).round().astype(int)
    assert len(np.unique(pick)) == N_TILES, len(tiles)
    lib = np.stack([tiles[i] for i in pick])
    sheet = lib.reshape(SHEET_H, SHEET_W, TILE, TILE, 3).transpose(0, 2, 1, 3, 4).reshape(SHEET_H * TILE, SHEET_W * TILE, 3)
    target.save(os.path.join(HERE, "photo_pd_target.jpg"), quality=92, subsampling=0, optimize=True)
    Image.fromarray(sheet).save(os.path.join(HERE, "photo_pd_library.jpg"), quality=92, subsampling=0, optimize=True)
    t_bgr, lib_bgr = _decode()
    meta = dict(target=TARGET, target_size=[SIDE, SIDE], target_crop=list(box), library_from=OTHERS, tiles_available=len(tiles),
                tiles=N_TILES, tile=TILE, scales=[2, 3],
                resize="target: centred square, PIL Image.LANCZOS to 4096; library photographs PIL Image.BOX by 1/2 and 1/3",
                storage="JPEG quality 92, 4:4:4; the sha256 are of the arrays as DECODED from the committed files",
                pillow=Image.__version__ if hasattr(Image, "__version__") else None,
                target_sha256=hashlib.sha256(t_bgr.tobytes()).hexdigest(), library_sha256=hashlib.sha256(lib_bgr.tobytes()).hexdigest())
    json.dump(meta, open(os.path.join(HERE, "photo_pd_fixture.json"), "w"), indent=1)
    print(meta)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/SampleImages")
