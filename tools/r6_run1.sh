#!/bin/bash
# round 6, GPU call 1: the 4/8-rank plugin tests + host profile of the IMPALA plugin path
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
nproc > gpurun_out/r6a_box.txt; free -g >> gpurun_out/r6a_box.txt
( time timeout 1500 python -m pytest tests/test_gpu_dp_ranks.py -m gpu -q --durations=20 ) > gpurun_out/r6a_ranks.log 2>&1
tail -40 gpurun_out/r6a_ranks.log
timeout 300 python tools/impala_host_profile.py breakout_impala > gpurun_out/r6a_hostprof_breakout.log 2>&1
timeout 300 python tools/impala_host_profile.py pong_impala_speedup > gpurun_out/r6a_hostprof_pong.log 2>&1
head -45 gpurun_out/r6a_hostprof_breakout.log
