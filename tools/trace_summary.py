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
    everything = "--all" in sys.argv
    for row in csv.DictReader(open(tr[0])):
        if "xnrs::" not in row["Kernel_Name"]:
            if not everything:
                continue
            key = ("[other] " + row["Kernel_Name"].split("(")[0].split("<")[0][-50:], 0)  # torch / rccl kernels, by name
        else:
            key = (short(row["Kernel_Name"]), int(row.get("Grid_Size", row.get("Grid_Size_X", 0))))
        d[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[key] = (row.get("VGPR_Count", "?"), row.get("LDS_Block_Size", "?"))
    n_all = sum(len(v) for v in d.values())
    t_all = sum(sum(v) for v in d.values())
    n_small = sum(1 for v in d.values() for x in v if x < 30000)
    t_small = sum(x for v in d.values() for x in v if x < 30000)
    print(f"# launches {n_all}, kernel time {t_all / 1e6:.3f} ms; under 30 us: {n_small} launches, {t_small / 1e6:.3f} ms")
    print(f"{'kernel':62s} {'grid':>9s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0]:62s} {k[1]:9d} {meta[k][0]:>5s} {meta[k][1]:>6s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
