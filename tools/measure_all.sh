#!/bin/bash
# On the GPU box: the measurement pass behind the numbers of DESIGN.md / README.md / bench.py's roofline.traffic.
#   gpurun -- tools/measure_all.sh <tag>      (results under gpurun_out/<tag>/; copy what is cited into profiles/)
# 1. PMC passes (HBM traffic, VALU / LDS instruction counts) per workload -> traffic_<workload>.json
# 2. rocprofv3 --kernel-trace --stats of the cfg3 bench command -> *_kernel_trace_stats.txt
# 3. the default bench.py line (headline + secondaries + CPU baselines)
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for wl in cfg3 cfg2 cfg5 cfg4 cfg1 cfg4d cfg4s cfg3u cfgmix; do
  timeout 900 python tools/measure_traffic.py --workload $wl --out $out/traffic_$wl.json > $out/traffic_$wl.log 2>&1 || echo "traffic $wl FAILED"
done
for wl in cfg3 cfg5 cfg4 cfg4d cfg4s cfgmix; do
  rm -rf $out/prof_$wl
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_$wl -o $wl -- python bench.py --workload $wl --no-cpu-baseline --no-secondary > $out/prof_$wl.log 2>&1
  db=$(find $out/prof_$wl -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > $out/${wl}_kernel_trace_stats.txt
  # the database behind the summary stays (under gpurun_out/, not committed) so that the figures can be re-aggregated
  [ -n "$db" ] && cp $db $out/${wl}_results.db
  rm -rf $out/prof_$wl
done
# the bench line quotes a PMC pass only for the device code it was taken on (source_sha): put this pass in place first
cp $out/traffic_cfg*.json profiles/
HFCL_BENCH_FULL_DIR=$out timeout 1500 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
tail -1 $out/bench_default.json | cut -c1-400
