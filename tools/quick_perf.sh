#!/bin/bash
# Fast GPU iteration: the kernel-facing parity tests, per-layer kernel times, the headline number.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_learner.py -x -q -m gpu -k "${1:-step or layer or impala or train}" 2>&1 | tail -3
python tools/layer_bench.py fused 2>&1 | tail -14
python bench.py --quick --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PPO', round(d['value']), d['ms_per_step'], d['roofline']["kernels_us_isolated"])"
python bench.py --workload breakout_impala --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('IMP', round(d['value']), d['us_per_train'], d['roofline']["kernels_us_isolated"])"
