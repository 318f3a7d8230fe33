#!/bin/bash
# round 4: counter traffic of the dominant kernels from THIS binary (Q/K/V projection, gather-only kernel, StandardRec's fc1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_traffic.sh r04t 2>&1 | tail -n 8
PMC_SETS="fetch write" COLS=200 bash tools/gpu_pmc_any.sh r04std prof_other_models.py standard 3 2>&1 | tail -n 12
bash tools/gpu_gather.sh r04g 2>&1 | tail -n 20
# keep what traffic_json.py reads, drop the rest (merge limit)
find gpurun_out/prof_r04t gpurun_out/prof_r04g gpurun_out/prof_r04std -type f ! -name '*counter_collection.csv' ! -name 'summary.txt' -size +200k -delete
du -sh gpurun_out/prof_r04t gpurun_out/prof_r04g gpurun_out/prof_r04std
