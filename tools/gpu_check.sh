#!/bin/bash
# One GPU-box call: the -m gpu suite, the default bench line, and in-graph kernel averages of the three workloads.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests -m "not gpu" -x -q > gpurun_out/${TAG}_pytest_cpu.log 2>&1; tail -2 gpurun_out/${TAG}_pytest_cpu.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -3 gpurun_out/${TAG}_bench.err
python - <<P
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("PPO value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"])
    print("kernels", d["roofline"]["kernels_us_isolated"])
    print("sustained", d.get("sustained"))
    for k, v in d.get("e2e", {}).items():
        if isinstance(v, dict): print("e2e", k, {q: (round(x, 3) if isinstance(x, float) else x) for q, x in v.items() if q != "path"})
    for s in d.get("secondary", []):
        print("SEC", s["workload"][:40], "value", round(s["value"]), "us/train", round(s["us_per_train"], 1), "frac", round(s["update_frac_of_fp32_mfma_peak"], 3), "e2e", {q: (round(x, 3) if isinstance(x, float) else x) for q, x in s["e2e"].items() if q != "path"}, "publish", {q: (round(x, 3) if isinstance(x, float) else x) for q, x in s.get("e2e_publish", {}).items() if q != "path"}, "async", {q: (round(x, 3) if isinstance(x, float) else x) for q, x in s.get("e2e_publish_async_loss", {}).items() if q not in ("path", "note")})
        print("   kernels", s["roofline"]["kernels_us_isolated"], "cpu", s.get("cpu_baseline", {}).get("value"))
    print("cpu", d.get("cpu_baseline"))
    print("library", d.get("library"), "box", d.get("box"))
    ms = d.get("modelled_scaling", {})
    print("modelled step us", ms.get("measured_sgd_step_us_by_rows"), "strict8", ms.get("strict", {}).get("8"), "weak8", ms.get("weak", {}).get("8"))
except Exception as e:
    print("bench parse failed", e)
P
cd /tmp && export TMPDIR=/tmp
for W in ppo breakout_impala pong_impala_speedup; do
  rm -rf /tmp/qs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -- python $R/bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --quick > /tmp/qs.log 2>&1
  f=$(find /tmp/qs -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/${TAG}_kernel_stats_$W.csv
  python - <<P
import csv
print("---- $W")
for r in csv.DictReader(open("$f")):
    if int(r["Calls"]) > 50: print("%6d %8.2f us  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
P
done
