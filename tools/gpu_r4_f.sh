#!/bin/bash
# round 4, call f: where the N = 256 product's bytes go (PMC) and the BK 32 / 2-workgroup variant beside the shipped one
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
PMC_SETS="fetch tcc" COLS=300 bash tools/gpu_pmc_any.sh r04n256 probes/n256_traffic.py > gpurun_out/r04n256_pmc.txt 2>&1
tail -40 gpurun_out/r04n256_pmc.txt
echo "== shipped (BK 16, 4 WG/CU)"; python3 tools/bench_n256.py 2>&1 | tee gpurun_out/r04n256_default.txt
echo "== PIPE 5 BK 32 (2 WG/CU)"; XNRS_GEMM_PIPE=5 python3 tools/bench_n256.py 2>&1 | tee gpurun_out/r04n256_bk32.txt
