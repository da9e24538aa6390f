#!/usr/bin/env python3
"""Turns the rocprofv3 passes of tools/profile.sh (all of the SAME command,
`python3 bench.py --config X --steps 1 --warmup 3 --no-cpu` (warm-up steps first: the item size of a CIEDE2000
launch can follow the engine's previous launch, so the counted launches are steady-state ones), separate runs as the MI355X guide
prescribes: --pmc FETCH_SIZE, --pmc WRITE_SIZE, --pmc SQ_*) into the per-launch record of the
dominant kernel that bench.py prices its roofline with (profiles/pmc_traffic.json).

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide
coalesced streaming reads, i.e. reports HALF the bytes -> doubled here.  WRITE_SIZE is taken as
is.  Both counters are in KiB.  SQ_* counters are summed over the launch as rocprofv3 reports
them (SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* in quad-cycles, SQ_INSTS_VALU in wave-instructions).

The record carries a hash of the kernel sources and the engine options of the run; bench.py
refuses a record that does not match the build it is timing.

usage: summarise_pmc.py CONFIG DIR [--opt NAME=VALUE ...]
       DIR holds fetch/, write/, sq/ (rocprofv3 -d) and bench_sq.json (the bench line of the SQ pass)
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


FILTER = r"\bfilter_kernel<|filter_lds_kernel<|mm::score_kernel<"


def per_kernel(dirname):
    """{kernel name: {counter: (launches, sum)}} from a rocprofv3 counter_collection.csv under dirname -- of the dominant
    (filter) kernel only the LAST dispatch: a large job measures the fp16 pre-tier over its first two launches (two
    instantiations of filter_kernel), the counted step is the one after them."""
    rows = []
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(path)))
    filters = [r for r in rows if re.search(FILTER, r["Kernel_Name"])]
    # (the MAIN launch of the last step: deepening rounds launch the filter again over a handful of cells -- far smaller grids)
    biggest = max((int(r["Grid_Size"]) for r in filters), default=0)
    last = max((int(r["Dispatch_Id"]) for r in filters if 2 * int(r["Grid_Size"]) >= biggest), default=None)
    out = {}
    for r in rows:
        if re.search(FILTER, r["Kernel_Name"]) and int(r["Dispatch_Id"]) != last:
            continue
        k = out.setdefault(r["Kernel_Name"], {})
        n, v = k.get(r["Counter_Name"], (0, 0.0))
        k[r["Counter_Name"]] = (n + 1, v + float(r["Counter_Value"]))
    return out


def dominant(table):
    return next((k for k in table if re.search(r"\bfilter_kernel<", k)), None) or \
        next((k for k in table if re.search(r"filter_lds_kernel<", k)), None) or \
        next(k for k in table if re.search(r"mm::score_kernel<", k))


def main():
    config, d = sys.argv[1:3]
    opts = [a for a in sys.argv[3:] if a != "--opt"]
    import bench
    fetch, write, sq = per_kernel(os.path.join(d, "fetch")), per_kernel(os.path.join(d, "write")), per_kernel(os.path.join(d, "sq"))
    name = dominant(fetch)
    fn, fv = fetch[name]["FETCH_SIZE"]
    wn, wv = write[name]["WRITE_SIZE"]
    entry = {
        "kernel": re.search(r"(\w+_kernel)<", name).group(1),
        "kernel_src_sha16": bench.kernel_source_hash(),
        "options": opts,
        "launches": fn,
        "FETCH_SIZE_KiB_per_launch": fv / fn,
        "WRITE_SIZE_KiB_per_launch": wv / wn,
        "fetch_correction": 2.0,
        "traffic_bytes_per_launch": (2.0 * fv / fn + wv / wn) * 1024.0,
        "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_*> -- python3 bench.py --config %s --steps 1 --warmup 3 --no-cpu %s" % (
            config, " ".join("--opt " + o for o in opts)),
    }
    if name in sq:
        for counter, (n, v) in sorted(sq[name].items()):
            entry[counter + "_per_launch"] = v / n
    bench_sq = os.path.join(d, "bench_sq.json")
    if os.path.exists(bench_sq):
        line = [ln for ln in open(bench_sq).read().splitlines() if ln.startswith("{")][-1]
        doc = json.loads(line)
        entry["visited_groups_per_launch"] = doc["roofline"]["visited_groups_per_launch"]
        entry["avg_launch_ms_under_pmc"] = doc["roofline"]["avg_launch_ms"]
        entry["filter_tile"] = doc["last_step_device_ms"].get("filter_tile")     # candidates per work item the passes ran with
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[config] = entry
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
