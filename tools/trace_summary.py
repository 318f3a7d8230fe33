#!/usr/bin/env python
"""Per (kernel, grid) duration table from a rocprofv3 kernel trace CSV (the file behind profiles/*_kernel_trace_summary.txt).

    python tools/trace_summary.py gpurun_out/prof_<tag>/bench_trace > profiles/rNN_bench_kernel_trace_summary.txt
"""
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
    return name


def main(root):
    tr = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    d = defaultdict(list)
    meta = {}
    for row in csv.DictReader(open(tr[0])):
        if "xnrs::" not in row["Kernel_Name"]:
            continue
        key = (short(row["Kernel_Name"]), int(row.get("Grid_Size", row.get("Grid_Size_X", 0))))
        d[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[key] = (row.get("VGPR_Count", "?"), row.get("LDS_Block_Size", "?"))
    print(f"{'kernel':62s} {'grid':>9s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0]:62s} {k[1]:9d} {meta[k][0]:>5s} {meta[k][1]:>6s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
