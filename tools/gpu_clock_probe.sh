#!/bin/bash
# Samples rocm-smi (clocks, power) every ~0.25 s beside tools/probes/clock_probe.py and prints the mean per GEMM window.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/r04_clock_probe
mkdir -p $OUT
rocm-smi --showclocks --showpower > $OUT/smi_once.txt 2>&1; head -n 30 $OUT/smi_once.txt
( while true; do echo "T $(date +%s.%N)"; rocm-smi --showclocks --showpower --csv 2>/dev/null; sleep 0.2; done ) > $OUT/smi.log 2>&1 &
POLL=$!
timeout -k 10 120 python3 tools/probes/clock_probe.py 4 2>&1 | grep -v amdgpu.ids | tee $OUT/windows.txt
kill $POLL
python3 - <<'PY'
import re
out='gpurun_out/r04_clock_probe'
wins=[l.split() for l in open(out+'/windows.txt') if l.startswith('WINDOW')]
samples=[]; t=None
for l in open(out+'/smi.log'):
    if l.startswith('T '): t=float(l.split()[1]); continue
    if l.startswith('card0') or l.startswith('0,'):
        samples.append((t,l.strip()))
hdr=[l.strip() for l in open(out+'/smi.log') if l.startswith('device')][:1]
print('header', hdr)
for w in wins:
    a,b=float(w[1]),float(w[2])
    rows=[s for (ts,s) in samples if a+1.5<=ts<=b]
    print(' '.join(w[3:]), 'samples', len(rows))
    for r in rows[:3]+rows[-2:]: print('   ', r[:200])
PY
