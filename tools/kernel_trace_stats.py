#!/usr/bin/env python
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace run WITH median, p99 and an outlier count (VERDICT r5 item
6b: rocprofv3's own --stats table has mean / min / max only, and one 19.9 ms launch in 211 turned a 19.4 us kernel into a
"113.9 us" one in profiles/r05_kernel_stats_pong_impala_speedup.csv without anybody noticing).

    python tools/kernel_trace_stats.py <rocprofv3 output dir> <out.csv>

Reads every *kernel_trace.csv below the directory (columns Kernel_Name, Start_Timestamp, End_Timestamp), writes one row per
kernel: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev (the columns of rocprofv3's
kernel_stats.csv, so bench.py / tools/pmc_summary.py read it unchanged) + MedianNs, P99Ns, Outliers (launches longer than
10 x the median) + TrimmedAverageNs (mean without the outliers).  Prints the kernels that have outliers."""
import csv
import glob
import os
import sys

import numpy as np


def main():
    src, out = sys.argv[1], sys.argv[2]
    per = {}
    for path in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                per.setdefault(row["Kernel_Name"], []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    if not per:
        raise SystemExit("no *kernel_trace.csv with rows below {}".format(src))
    total = float(sum(sum(v) for v in per.values()))
    rows = []
    for name, v in per.items():
        a = np.asarray(v, np.float64)
        med = float(np.median(a))
        outl = a > 10.0 * med
        rows.append({"Name": name, "Calls": len(v), "TotalDurationNs": int(a.sum()), "AverageNs": float(a.mean()),
                     "Percentage": 100.0 * a.sum() / total, "MinNs": int(a.min()), "MaxNs": int(a.max()), "StdDev": float(a.std()),
                     "MedianNs": med, "P99Ns": float(np.percentile(a, 99)), "Outliers": int(outl.sum()),
                     "TrimmedAverageNs": float(a[~outl].mean())})
    rows.sort(key=lambda r: -r["TotalDurationNs"])
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        if r["Outliers"]:
            print("OUTLIERS: {} of {} launches of {} longer than 10 x the median {:.1f} us (max {:.1f} us; mean {:.1f} -> {:.1f} us "
                  "without them)".format(r["Outliers"], r["Calls"], r["Name"][:70], r["MedianNs"] / 1e3, r["MaxNs"] / 1e3,
                                         r["AverageNs"] / 1e3, r["TrimmedAverageNs"] / 1e3))


if __name__ == "__main__":
    main()
