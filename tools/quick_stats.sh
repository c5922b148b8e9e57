#!/bin/bash
# In-graph per-kernel averages of a short bench run (rocprofv3 --kernel-trace --stats) + the bench value; GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /tmp/qs.log 2>&1
python - <<'P'
import csv,glob
f=glob.glob("/tmp/qs/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "xt::" in r["Name"] and int(r["Calls"])>100: print("%6d %8.2f us  %s"%(int(r["Calls"]),float(r["AverageNs"])/1e3,r["Name"][:90]))
P
grep '^{"metric"' /tmp/qs.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
