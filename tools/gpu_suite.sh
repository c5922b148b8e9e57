#!/bin/bash
# round 6, GPU call: the whole -m gpu suite (no -x) + the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r6b}
cd $R
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q --durations=15 ${2:+-k "$2"} ) > gpurun_out/${TAG}_pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest.log | tail -40
if [ -z "$3" ]; then
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -3 gpurun_out/${TAG}_bench.err
cp bench_detail.json gpurun_out/${TAG}_bench_detail.json 2>/dev/null
python - <<P
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("PPO value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "e2e", d.get("value_e2e"))
    print("cpu", d.get("cpu_baseline"))
    for k in ("secondary", "actor_scan", "modelled_scaling"):
        print(k, json.dumps(d.get(k))[:600])
except Exception as e:
    print("bench parse failed", e)
P
fi
