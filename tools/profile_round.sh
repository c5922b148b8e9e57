#!/bin/bash
# One GPU-box call that refreshes the judged profile artifacts (run from the repo root on the GPU box).  Counter
# passes are separate rocprofv3 runs with --pmc only (no trace domains), as MI355X_MICROARCH.md prescribes.
#   gpurun_out/<tag>_kernel_stats_<workload>.csv rocprofv3 --kernel-trace --stats of bench.py --quick (in-graph averages)
#   gpurun_out/<tag>_pmc_<workload>.json         tools/pmc_summary.py of three --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*) over
#                                                tools/step_probe.py <workload> (eager updates: every kernel of a step)
#   gpurun_out/<tag>_bench.json                 the default `python bench.py` line (the compact driver-facing one), run LAST so that
#   gpurun_out/<tag>_bench_detail.json          its roofline joins the counters just collected (copied to profiles/ on the box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for W in ppo breakout_impala pong_impala_speedup; do
  rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --quick > /tmp/ks.log 2>&1
  # per-kernel mean / MEDIAN / p99 / outlier count from the raw trace (rocprofv3's own stats table has no median)
  python $R/tools/kernel_trace_stats.py /tmp/ks $R/gpurun_out/${TAG}_kernel_stats_$W.csv || cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats_$W.csv
done
for W in ppo breakout_impala pong_impala_speedup; do
  for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    set -- $P; name=$1; shift
    rm -rf /tmp/pmc_${W}_$name
    rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_${W}_$name -- python $R/tools/step_probe.py $W > /tmp/pmc_${W}_$name.log 2>&1
  done
  (cd $R && python tools/pmc_summary.py /tmp/pmc_${W}_fetch /tmp/pmc_${W}_write /tmp/pmc_${W}_sq gpurun_out/${TAG}_pmc_$W.json gpurun_out/${TAG}_kernel_stats_$W.csv | tail -16)
  cp $R/gpurun_out/${TAG}_pmc_$W.json $R/profiles/${TAG}_pmc_$W.json
done
cd $R
python $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
cp $R/bench_detail.json $R/gpurun_out/${TAG}_bench_detail.json
wc -c gpurun_out/${TAG}_bench.json
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-1500
