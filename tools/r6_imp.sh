#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for W in breakout_impala pong_impala_speedup; do
python bench.py --workload $W --no-cpu-baseline --no-in-graph-stats --detail-file /tmp/dd.json > /tmp/oo.json 2>/tmp/ee.log || tail -5 /tmp/ee.log
python - <<P
import json
d = json.loads(open("/tmp/oo.json").read().strip().splitlines()[-1])
print("$W", round(d["value"]), "us/train", round(d["us_per_train"],1))
for k in ("e2e_publish", "e2e_ring_prefetch", "e2e_ring_blocking"):
    v = d.get(k) or {}
    print("   ", k, {q: (round(x, 3) if isinstance(x, float) else x) for q, x in v.items() if q not in ("path", "note", "unit", "served_min_max")})
P
done
python -m pytest tests/test_gpu_prefetch.py -q -m gpu 2>&1 | tail -2
