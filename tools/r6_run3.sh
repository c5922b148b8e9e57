#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for P in 1 2 4 8; do timeout 300 python tools/dp_tail_probe.py --procs $P --rows 40 2>&1 | grep "^{" ; done | tee gpurun_out/r6c_tail_probe.txt
timeout 300 python tools/dp_tail_probe.py --procs 1 --rows 320 2>&1 | grep "^{" | tee -a gpurun_out/r6c_tail_probe.txt
( time timeout 1500 python -m pytest tests -m gpu -q -k "dp_fused or dp_ranks or tail or pixel_control or preflight" ) > gpurun_out/r6c_pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r6c_pytest.log | tail
