cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -n 8
python bench.py --no-cpu-baseline > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err; tail -c 300 gpurun_out/r4_bench2.err
python3 - <<PY
import json
d=json.load(open('gpurun_out/r4_bench2.json'))
print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'roofline', round(d['roofline']['frac'],3))
e=d['extra']
print({k: round(v,3) for k,v in e['stage_ms_per_step'].items() if v})
for k in ('nrms_train_step_B64','standard_train_step_B64','naml_train_step_B64'):
    t=e[k]; print(k, round(t['ms'],2))
print('news_only', {k: (round(v['ms'],3), round(v['frac_fp32_mfma'],3)) for k,v in e['news_encoder_only_1024'].items()})
print('latency', e['latency_one_impression'])
PY
