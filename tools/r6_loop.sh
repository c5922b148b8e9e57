#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3 4 5 6; do
  python bench.py --gpus 2 --test-backend gloo --steps 2 --warmup 1 --dp-variants ${1:-all} --detail-file /tmp/d$i.json > /tmp/o$i.json 2> /tmp/e$i.log
  rc=$?
  python - <<P
import json
try:
    d=json.load(open("/tmp/d$i.json")); v=d.get("dp_variants",{}).get("direct")
    print("run $i rc=$rc direct:", json.dumps(v)[:200])
except Exception as e:
    print("run $i rc=$rc no detail", e)
P
  grep -i "HSA\|fault\|abort\|terminate\|violation\|core dump\|what()" /tmp/e$i.log | grep -v "detail:" | cut -c1-300 | head -8
done
