"""Per step of a bench run under `rocprofv3 --kernel-trace`: the sum of the kernels' durations, the span from the first kernel's start to
the last one's end, and the idle time between consecutive kernels -- what a captured graph could remove at most on the device side.
Steps are told apart by the kernel that opens one (result_init_kernel)."""
import csv
import glob
import os
import sys

root = sys.argv[1]
for cfg in ("H", "C", "B"):
    paths = glob.glob(os.path.join(root, "kt_" + cfg, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        continue
    rows = list(csv.DictReader(open(paths[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], []
    for r in rows:
        if "result_init_kernel" in r["Kernel_Name"] and cur:
            steps.append(cur); cur = []
        cur.append(r)
    steps.append(cur)
    steps = [s for s in steps if any("result_init_kernel" in r["Kernel_Name"] for r in s)][-20:]      # the timed steps
    steps = [[r for r in s if "stream_read" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]] for s in steps]   # (the calibration after the last step)
    busy = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s) for s in steps]
    span = [int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in steps]
    idle = [sp - b for sp, b in zip(span, busy)]
    period = [int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"]) for a, b in zip(steps, steps[1:])]
    n = [len(s) for s in steps]
    med = lambda v: sorted(v)[len(v) // 2] / 1e3
    print("config %s: %d kernels per step; kernels busy %.1f us, first start -> last end %.1f us (idle between kernels %.1f us), step period %.1f us (host turnaround %.1f us)"
          % (cfg, sorted(n)[len(n) // 2], med(busy), med(span), med(idle), med(period), med(period) - med(span)))
    name = lambda r: r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    last = steps[-2]
    for a, b in zip(last, last[1:]):
        gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        print("    %-44s %8.1f us   then idle %6.1f us" % (name(a)[-44:], (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3, gap))
    print("    %-44s %8.1f us" % (name(last[-1])[-44:], (int(last[-1]["End_Timestamp"]) - int(last[-1]["Start_Timestamp"])) / 1e3))
