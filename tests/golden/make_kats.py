#!/usr/bin/env python3
"""Extracts the colour-difference known-answer vectors the reference's own tests
hold (test/tst_ColourDifference.h: RGBEuclidean :26-35, CIE76 :49-56, CIEDE2000
:73-108, the 34 Sharma-2005 pairs) into colour_diff_kats.json.

Runs ONLY in the build container, where /root/reference exists; the JSON it
writes is the committed fixture (data: inputs + expected outputs + tolerance).
"""
import json
import os
import re
import sys

REF = "/root/reference/test/tst_ColourDifference.h"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "colour_diff_kats.json")

# name of the gtest case -> (diff type id of ColourDifference::Type, tolerance in the test)
CASES = {"RGBEuclidean": (0, 1e-8), "CIE76": (1, 1e-8), "CIEDE2000": (2, 1e-4)}

NUM = r"[-+]?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?"
ROW = re.compile(
    r"\{\{\s*(%s)\s*,\s*(%s)\s*,\s*(%s)\s*\}\s*,\s*\{\s*(%s)\s*,\s*(%s)\s*,\s*(%s)\s*\}\s*,\s*(%s)\s*\}"
    % ((NUM,) * 7))


def main():
    text = open(REF, encoding="utf-8", errors="replace").read()
    out = {"source": "test/tst_ColourDifference.h (reference v3.4.2)", "cases": {}}
    for name, (type_id, tol) in CASES.items():
        m = re.search(r"TEST\(ColourDifference,\s*%s\)(.*?)\n\}" % name, text, re.S)
        if not m:
            sys.exit("case %s not found" % name)
        rows = [[float(v) for v in r] for r in ROW.findall(m.group(1))]
        out["cases"][name] = {
            "type": type_id,
            "tolerance": tol,
            "pairs": [{"first": r[0:3], "second": r[3:6], "difference": r[6]} for r in rows],
        }
        print(name, len(rows), "pairs")
    json.dump(out, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
