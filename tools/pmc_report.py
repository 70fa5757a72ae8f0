#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel and counter, the mean over dispatches of the per-dispatch total.

    python tools/pmc_report.py gpurun_out/pmc_dir [more dirs] [--kernel substring] [--csv out.csv]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    args = sys.argv[1:]
    kfilter, out = None, None
    dirs = []
    while args:
        a = args.pop(0)
        if a == "--kernel":
            kfilter = args.pop(0)
        elif a == "--csv":
            out = args.pop(0)
        else:
            dirs.append(a)
    acc = defaultdict(lambda: defaultdict(float))  # (kernel, counter) -> dispatch -> value
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = r["Kernel_Name"]
                    if kfilter and kfilter not in k:
                        continue
                    acc[(k, r["Counter_Name"])][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    rows = []
    for (k, c), disp in sorted(acc.items()):
        vals = list(disp.values())
        rows.append((c, k, len(vals), sum(vals) / len(vals)))
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
    w.writerow(["counter", "kernel", "dispatches", "mean_value"])
    for r in rows:
        w.writerow([r[0], r[1][:120], r[2], f"{r[3]:.6g}"])


if __name__ == "__main__":
    main()
