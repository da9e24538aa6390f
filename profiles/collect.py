#!/usr/bin/env python3
"""Copies what tools/profile.sh left under gpurun_out/<tag>/ into profiles/ as <tag>_*: the PMC record
(pmc_traffic.json), the kernel-trace summary, the PMC dumps of configs D / C / B trimmed to the engine's own
kernels, the bench lines of every configuration and the filter's resource-usage remarks.
usage: python profiles/collect.py r02_v17"""
import csv
import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KEEP = ("filter_kernel", "filter_lds_kernel", "rescore_kernel", "seed_score_kernel", "seed_kernel", "contenders_kernel", "resolve_wave", "score_kernel")


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(HERE, "pmc_traffic.json"))
    for name, dst in (("kt/d_kernel_stats.csv", "_configD_kernel_stats.csv"), ("kt_PD/d_kernel_stats.csv", "_configPD_kernel_stats.csv"),
                      ("filter_resource_usage.txt", "_filter_resource_usage.txt")):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(HERE, tag + dst))
    for c in ("D", "C", "B", "E8", "E", "A", "F", "G", "H", "I", "P", "PD"):
        if not os.path.isdir(os.path.join(src, "pmc_" + c)):
            continue
        for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("SQ", "sq")):
            rows = list(csv.reader(open(os.path.join(src, "pmc_" + c, sub, "p_counter_collection.csv"))))
            keep = [rows[0]] + [r for r in rows[1:] if any(k in r[8] for k in KEEP)]
            csv.writer(open(os.path.join(HERE, "%s_config%s_pmc_%s.csv" % (tag, c, counter)), "w", newline="")).writerows(keep)
    for path in glob.glob(os.path.join(src, "bench_*.json")):
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        if lines:
            name = os.path.basename(path)[len("bench_"):-len(".json")]
            open(os.path.join(HERE, "%s_%s_bench.json" % (tag, name)), "w").write(lines[-1] + "\n")
    print("collected", tag)


if __name__ == "__main__":
    main()
