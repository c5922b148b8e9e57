#!/bin/bash
# In-graph per-kernel averages of a short headline run (rocprofv3 --kernel-trace --stats) + the bench value; GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -- python $R/bench.py --steps 8 --warmup 2 --quick --no-cpu-baseline > /tmp/qs.log 2>&1
python - <<'P'
import csv,glob
f=glob.glob("/tmp/qs/**/*kernel_stats.csv",recursive=True)[0]
tot=0.0
for r in csv.DictReader(open(f)):
    if "xt::" in r["Name"] and int(r["Calls"])>100:
        print("%6d %8.2f us  %s"%(int(r["Calls"]),float(r["AverageNs"])/1e3,r["Name"][:90])); tot+=float(r["AverageNs"])/1e3*(2 if "igemm_fwd_kernel<64, 64, 2, 2, false, false, 2" in r["Name"] else 1)
print("sum of in-graph averages per SGD step: %.1f us"%tot)
P
grep '^{"metric"' /tmp/qs.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
for i in 1 2 3; do python $R/bench.py --quick --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unprofiled', round(d['value']), d['ms_per_step'])"; done
