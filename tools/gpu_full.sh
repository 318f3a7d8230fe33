#!/bin/bash
# Runs on the GPU box (via gpurun): the whole -m gpu suite, smoke(), and the default bench line.  usage: tools/gpu_full.sh <tag>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-full}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 800 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python3 - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'whole', d['whole_path']['frac_fp32_mfma'], 'parity', d.get('parity_max_rel_err_vs_cpu'))
e=d['extra']
print('stages', e['stage_ms_per_step']); print('stage_tf', {k: round(v,1) for k,v in e['stage_tflops'].items()})
for k,v in e['other_models_fwd_B512_H25'].items(): print(k, round(v['impressions_per_s']), round(v['frac_fp32_mfma'],3), 'fc1', round(v['roofline']['achieved'],1), v['stage_ms_per_step'], v['parity_max_rel_err_vs_cpu'])
print('news_only', {k: (round(v['ms'],3), round(v['frac_fp32_mfma'],3)) for k,v in e['news_encoder_only_1024'].items()})
for k in ('nrms_train_step_B64','standard_train_step_B64'):
    t=e[k]; print(k, round(t['ms'],2), 'roof', round(t['roofline']['frac'],3), 'dW', round(t['roofline']['dominant_kernel']['achieved'],1), t['roofline']['stage_ms_profiled_step'])
print('upload', e['store_file_to_hbm']); print('eval', e['eval_epoch']['seconds'], 'idpath', e['id_path_B512']['gather']['ms'], e['id_path_B512']['gather_dedup']['ms'])
print('padfree', {k: round(v['ms_per_step'],1) for k,v in e['padding_free'].items() if isinstance(v, dict)})
PY
