#!/bin/bash
# Round-4 evidence run on the GPU box: full GPU suite, smoke, the default bench line, its kernel trace, and the grad-step
# traces (NRMS / StandardRec / NAML) + PMC passes of the NRMS grad step.  Summaries are copied to profiles/ by hand afterwards.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -n 2 $OUT/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout -k 10 300 python3 tools/soak_determinism.py 25 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_determinism.txt
timeout -k 10 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"
for t in nrms standard naml; do
  timeout -k 10 300 python bench.py --train $t --steps 20 --warmup 5 >> $OUT/bench_train_lines.json 2>> $OUT/bench.err
done
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py --no-extra --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_trace.err
python3 $R/tools/trace_summary.py $OUT/bench_trace > $OUT/bench_kernel_trace_summary.txt 2>&1
for m in nrms standard naml; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$m -- python3 $R/tools/prof_train.py 6 $m > $OUT/trace_$m.log 2>&1 || { echo "trace $m failed"; tail -n 3 $OUT/trace_$m.log; }
  python3 $R/tools/trace_summary.py $OUT/trace_$m --all > $OUT/train_step_${m}_kernel_trace_summary.txt 2>&1
done
mkdir -p $OUT/pmc
for pass in "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
            "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc/pmc_$name -- python3 $R/tools/prof_train.py 2 nrms > $OUT/pmc/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -n 3 $OUT/pmc/pmc_$name.log; }
done
python3 $R/tools/summarize_prof.py $OUT/pmc > $OUT/train_step_pmc.txt 2>&1
# keep the merged-back directory small: the raw traces stay on the box
rm -rf $OUT/bench_trace $OUT/trace_nrms $OUT/trace_standard $OUT/trace_naml $OUT/pmc
cd $R
python3 - <<PY
import json
d=json.load(open('$OUT/bench_line.json'))
print('value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'roofline', round(d['roofline']['frac'],3), 'parity', d.get('parity_max_rel_err_vs_cpu'), 'build', d['build_id'], d['build_is_tree'])
e=d['extra']
for k in ('nrms_train_step_B64','standard_train_step_B64','naml_train_step_B64'):
    t=e[k]; print(k, round(t['ms'],2), 'roof', round(t['roofline']['frac'],3), {kk: round(vv['ms'],2) for kk,vv in t.items() if isinstance(vv, dict) and 'ms' in vv})
print('ig', {k: round(v['it_per_s']) for k,v in e['ig_step'].items() if 'it_per_s' in v})
print('news_only', {k: (round(v['ms'],3), round(v['frac_fp32_mfma'],3)) for k,v in e['news_encoder_only_1024'].items()})
for k,v in e['other_models_fwd_B512_H25'].items(): print(k, round(v['impressions_per_s']), round(v['frac_fp32_mfma'],3))
PY
head -n 4 $OUT/train_step_nrms_kernel_trace_summary.txt | cut -c1-160
grep -h '"value"' $OUT/bench_train_lines.json | cut -c1-260
