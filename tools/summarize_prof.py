#!/usr/bin/env python
"""Summarise a tools/gpu_profile.sh output directory: per-kernel duration stats from the kernel
trace and per-kernel mean counter values from each PMC pass (CSV files written by rocprofv3)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for pre in ("void xnrs::", "xnrs::"):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:66]


def main(root):
    tr = glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        d = defaultdict(list)
        for row in csv.DictReader(open(tr[0])):
            key = short(row["Kernel_Name"]) + " grid=" + str(row.get("Grid_Size", row.get("Grid_Size_X", "?"))) + "x" + str(row.get("Grid_Size_Y", ""))
            d[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        tot = sum(sum(v) for v in d.values())
        print("== kernel trace (ns) : name calls total_ms avg_us min_us max_us pct")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            print(f"{k:86s} {len(v):5d} {sum(v)/1e6:9.3f} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f} {100*sum(v)/tot:5.1f}%")
    for pdir in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(pdir):
            continue
        cc = glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True)
        if not cc:
            print("==", os.path.basename(pdir), ": no counter csv")
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(cc[0])):
            key = short(row["Kernel_Name"]) + " grid=" + str(row.get("Grid_Size", "?"))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.basename(pdir), "(mean per dispatch)")
        for k, cs in acc.items():
            print(f"{k:86s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))


if __name__ == "__main__":
    main(sys.argv[1])
