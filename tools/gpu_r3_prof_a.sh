#!/bin/bash
# Round-3 profile run A (on the GPU box via gpurun): news-encoder pass + bench under rocprofv3 (tools/gpu_profile.sh) and
# the gather kernel's counters (tools/gpu_gather.sh).  Summaries are copied into profiles/ by tools/refresh_profiles.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_profile.sh r03 > gpurun_out/prof_r03_stdout.log 2>&1 && tail -3 gpurun_out/prof_r03_stdout.log | cut -c1-300 &&
bash tools/gpu_gather.sh r03g > gpurun_out/prof_r03g_stdout.log 2>&1 && tail -12 gpurun_out/prof_r03g_stdout.log | cut -c1-200
