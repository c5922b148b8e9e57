#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "one_launch_equals" 2>&1 | tail -5
python -m pytest tests/test_gpu_learner.py -q -m gpu -k "impala" 2>&1 | tail -3
python - <<'P'
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench
from xingtian_amd import lib as L
for knob in (0, 256, 0, 256):
    old = L.set_tuning(fwd_fuse12=knob)
    r = bench.bench_impala("breakout_impala", 10, 3, False, in_graph=False, quick=True)
    print("fwd_fuse12", knob, "us_per_train", round(r["us_per_train"], 2), "value", round(r["value"]))
    L.set_tuning(**old)
P
