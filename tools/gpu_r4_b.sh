set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_hip_train_step.py -x -q 2>&1 | tail -n 12 > gpurun_out/r4_t1.log
OUT=$R/gpurun_out/prof_r4train
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for m in nrms standard; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$m -- python3 $R/tools/prof_train.py 6 $m > $OUT/trace_$m.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace_$m.log; }
python3 $R/tools/trace_summary.py $OUT/trace_$m --all > $OUT/summary_$m.txt 2>&1
done
cd $R
tail -n 6 gpurun_out/r4_t1.log; head -n 50 $OUT/summary_nrms.txt | cut -c1-180
