#!/bin/bash
# round 3, first GPU pass: the new tests + kernel traces of StandardRec / NAML forward + a bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3a
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_naml_ids.py tests/test_hip_training.py tests/test_hip_news_fused.py tests/test_hip_data.py -x -q > gpurun_out/r3a/tests.log 2>&1
echo "tests rc=$?" ; tail -5 gpurun_out/r3a/tests.log
for m in standard NAML; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3a/trace_$m -- python3 $R/tools/prof_other_models.py $m 5 > $R/gpurun_out/r3a/trace_$m.log 2>&1)
  python3 tools/trace_summary.py gpurun_out/r3a/trace_$m > gpurun_out/r3a/trace_${m}_summary.txt 2>&1
  head -12 gpurun_out/r3a/trace_${m}_summary.txt | cut -c1-160
done
timeout -k 10 600 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3a/bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3a/bench.json'))
print(d['value'], d['roofline']['frac'])
print(json.dumps(d['extra']['other_models_fwd_B512_H25'], indent=0)[:1500])
print(d['extra']['news_encoder_only_1024'])
PY
