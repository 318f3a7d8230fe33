#!/bin/bash
# Runs on the GPU box (via gpurun): the gather stage under rocprofv3 -- kernel trace, then FETCH_SIZE and WRITE_SIZE in
# separate PMC passes (never combined with other trace domains).  usage: tools/gpu_gather.sh <tag>
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-gather}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
for dist in uniform zipf; do
  python3 $R/tools/prof_gather.py $dist 5 > $OUT/plain_$dist.log 2>&1 || { echo "plain $dist failed"; tail -5 $OUT/plain_$dist.log; }
  grep "^\[" $OUT/plain_$dist.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$dist -- python3 $R/tools/prof_gather.py $dist 3 > $OUT/trace_$dist.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace_$dist.log; }
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/${dist}_pmc_$name -- python3 $R/tools/prof_gather.py $dist 2 > $OUT/${dist}_pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $OUT/${dist}_pmc_$name.log; }
  done
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
root = "$OUT"
for dist in ("uniform", "zipf"):
    print("==", dist)
    tr = glob.glob(os.path.join(root, "trace_" + dist, "**", "*kernel_trace.csv"), recursive=True)
    dur = defaultdict(list)
    if tr:
        for row in csv.DictReader(open(tr[0])):
            k = row["Kernel_Name"]
            if "gather_rows" in k or "gemm_f32" in k:
                dur[k.split("(")[0][-70:] + " grid=" + row.get("Grid_Size", "?")].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in dur.items():
            print(f"  {k:90s} calls {len(v):3d} avg {sum(v)/len(v)/1e3:10.1f} us  min {min(v)/1e3:10.1f} us")
    for name in ("fetch", "write", "tcc"):
        cc = glob.glob(os.path.join(root, f"{dist}_pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
        if not cc:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(cc[0])):
            k = row["Kernel_Name"]
            if "gather_rows" in k or "gemm_f32" in k:
                acc[k.split("(")[0][-70:] + " grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            print(f"  {k:90s} " + "  ".join(f"{c}={sum(v)/len(v):.5g}" for c, v in sorted(cs.items())))
PY
