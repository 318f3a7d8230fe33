#!/bin/bash
# Round-3 profile run B: grad step (trace + PMC), StandardRec / NAML forward (trace + PMC), fused short-title kernel (PMC),
# stand-alone kernel benches (additive_fused, dW, news_fused sweep, device compaction).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r3prof
mkdir -p $O
LINES_=70 COLS=360 bash tools/gpu_pmc_train.sh r03t > $O/train.txt 2>&1; echo "train rc=$?"
for m in standard NAML; do
  bash tools/gpu_trace.sh r03_$m tools/prof_other_models.py $m 6 > $O/trace_$m.txt 2>&1
  PMC_SETS="sq lds" bash tools/gpu_pmc_any.sh r03_$m prof_other_models.py $m 4 > $O/pmc_$m.txt 2>&1; echo "$m rc=$?"
done
PMC_SETS="sq lds" bash tools/gpu_pmc_any.sh r03_nf prof_news.py 1024 8 30 320 16 > $O/pmc_nf.txt 2>&1; echo "nf rc=$?"
timeout -k 10 200 python tools/bench_af.py > $O/bench_af.txt 2>&1; echo "af rc=$?"
timeout -k 10 200 python tools/bench_dw.py > $O/bench_dw.txt 2>&1; echo "dw rc=$?"
timeout -k 10 300 python tools/bench_news_fused.py > $O/bench_nf.txt 2>&1; echo "nf bench rc=$?"
NF_SWEEP=1 timeout -k 10 300 python tools/bench_news_fused.py > $O/bench_nf_sweep.txt 2>&1; echo "nf sweep rc=$?"
timeout -k 10 200 python tools/bench_compact.py > $O/bench_compact.txt 2>&1; echo "compact rc=$?"
tail -n 4 $O/bench_af.txt; tail -n 4 $O/bench_dw.txt
