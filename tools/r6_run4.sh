#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -- python $R/tools/dp_tail_probe.py --procs 1 --rows 40 --repeats 1 > /tmp/qs.log 2>&1
f=$(find /tmp/qs -name "*kernel_stats.csv" | head -1)
python - <<P
import csv
for r in csv.DictReader(open("$f")):
    if int(r["Calls"]) > 50: print("%6d %8.2f us  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
P
