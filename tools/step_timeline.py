"""Median per-launch time of one replayed step from a rocprofv3 --kernel-trace csv. usage: step_timeline.py <r_kernel_trace.csv> [other.csv]
With two traces prints them side by side (A/B inside one gpurun call: numbers from different boxes differ by a few %)."""
import collections
import csv
import sys


def timeline(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "step_begin" in r["Kernel_Name"]]
    if len(idx) < 3:  # the piano-roll ends: the bookkeeping rides on the step's first launch, the two embedding GEMMs
        idx = [i for i, r in enumerate(rows) if "gemm_nt_pair_kernel" in r["Kernel_Name"]]
    acc = collections.defaultdict(list)
    n = idx[-1] - idx[-2]
    for a, b in zip(idx[10:-1], idx[11:]):
        if b - a != n:
            continue
        for k, r in enumerate(rows[a:b]):
            acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    names = [r["Kernel_Name"] for r in rows[idx[-2]:idx[-1]]]
    return names, [sorted(acc[k])[len(acc[k]) // 2] for k in range(n)]


def short(n):
    return n.replace("_ZN3mst", "").replace("void mst::", "").replace("mst::", "")[:44]


tl = [timeline(p) for p in sys.argv[1:3]]
for k, name in enumerate(tl[0][0]):
    cols = "  ".join(f"{t[1][k]:7.1f}" if k < len(t[1]) else "      -" for t in tl)
    d = f"  {tl[1][1][k] - tl[0][1][k]:+6.1f}" if len(tl) == 2 and k < len(tl[1][1]) else ""
    print(f"{k:3d} {short(name):44s} {cols}{d}")
print("total", "  ".join(f"{sum(t[1]):7.1f}" for t in tl))
