#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python $R/tools/impala_prefetch_probe.py ${1:-breakout_impala} prefetch ${2:-tail} > /tmp/tr.log 2>&1
tail -3 /tmp/tr.log | cut -c1-200
python - <<'P'
import csv, glob
ev = []
for path in glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40]))
for path in glob.glob("/tmp/tr/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    print("copy columns", list(rows[0].keys()) if rows else None)
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?") + " " + str(r.get("Bytes", r.get("Size", "")))))
ev.sort()
# find a steady-state window: last 400 events
t0 = ev[-260][0]
for s, e, n in ev[-260:-130]:
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
P
