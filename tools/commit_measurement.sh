#!/bin/bash
# Copy what a measurement pass (tools/measure_all.sh <tag> on the GPU box, then `python bench.py` with the traffic files in
# place -> gpurun_out/<tag2>/bench_default.json) produced into profiles/ under the round's names.
#   tools/commit_measurement.sh <tag of measure_all> <tag of the bench run> <round prefix, e.g. r03_z>
set -e
t1=$1; t2=$2; pre=$3
sha=$(python -c "import json;print(json.load(open('gpurun_out/$t1/traffic_cfg3.json'))['source_sha'])")
cp gpurun_out/$t1/traffic_cfg*.json profiles/
for wl in cfg3 cfg5 cfg4 cfg4d cfg4s cfgmix; do
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --no-cpu-baseline --no-secondary  (tools/measure_all.sh $t1, device code"
    echo "# source_sha $sha); the database is kept as gpurun_out/$t1/${wl}_results.db (scratch, not committed)"
    cat gpurun_out/$t1/${wl}_kernel_trace_stats.txt; } > profiles/${pre}_${wl}_kernel_trace_stats.txt
done
grep '^{' gpurun_out/$t2/bench_default.json | tail -1 > profiles/${pre}_bench_line.json   # the compact line (what the driver parses)
cp gpurun_out/$t2/bench_full.json profiles/${pre}_bench_default.json                      # the full record
echo "source_sha $sha -> profiles/${pre}_*"
