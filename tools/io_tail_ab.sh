#!/bin/bash
# same-box A/B of xt_train_io.tail_in_graph (model_config IO_TAIL_IN_GRAPH) on the ring-fed, prefetched IMPALA loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  for W in breakout_impala pong_impala_speedup; do
    for M in tail notail; do
      echo "== $W $M ($rep)"
      timeout 300 python tools/impala_prefetch_probe.py $W prefetch $M 2>&1 | grep -v amdgpu.ids | head -24
    done
  done
done
