#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for P in 1 2 4 8; do timeout 300 python tools/dp_tail_probe.py --procs $P --rows 40 2>&1 | grep "^{" ; done | tee gpurun_out/r6e_tail_probe.txt
bash tools/r6_loop.sh 2>&1 | grep "^run" | head -4
( time timeout 1800 python -m pytest tests -m gpu -q -k "dp_fused or dp_ranks or dp_plugin or test_gpu_dp or direct or prefetch or preflight" ) > gpurun_out/r6e_pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r6e_pytest.log | tail
