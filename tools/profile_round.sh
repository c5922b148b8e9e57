#!/bin/bash
# One GPU-box call that refreshes the judged profile artifacts (run from the repo root on the GPU box):
#   gpurun_out/prof_stats  rocprofv3 --kernel-trace --stats of `python bench.py`
#   gpurun_out/pmc_fetch, gpurun_out/pmc_write  separate --pmc passes over tools/layer_bench.py
#   gpurun_out/bench.json  the bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/tools/layer_bench.py fused > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/tools/layer_bench.py fused > $R/gpurun_out/pmc_write.log 2>&1
cd $R && python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic.json | tail -20
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -2
tail -1 gpurun_out/bench.json | cut -c1-200
