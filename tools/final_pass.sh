#!/bin/bash
# The round's last GPU call, under a deadline: the GPU test suite on the final code, then as many PMC passes / kernel traces as fit
# (most important first).  tools/final_pass.sh <tag> <seconds>
tag=${1:-fin}; deadline=${2:-800}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
left() { echo $(( deadline - SECONDS )); }
timeout 430 python -m pytest tests -q -m gpu -p no:cacheprovider < /dev/null > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
echo "after pytest: $(left) s left"
for wl in cfg3 cfg4s cfg4 cfg4d cfg5 cfg2 cfg1; do
  need=150; [ $wl = cfg4d ] && need=200; [ $wl = cfg5 ] && need=200
  if [ $(left) -lt $need ]; then echo "skip traffic $wl ($(left) s left)"; continue; fi
  timeout $need python tools/measure_traffic.py --workload $wl --out $out/traffic_$wl.json < /dev/null > $out/traffic_$wl.log 2>&1 || echo "traffic $wl FAILED"
  echo "traffic $wl done: $(left) s left"
  if [ $wl = cfg3 ] || [ $wl = cfg4s ]; then
    if [ $(left) -gt 120 ]; then
      rm -rf $out/prof_$wl
      timeout 110 rocprofv3 --kernel-trace --stats -d $out/prof_$wl -o $wl -- python bench.py --workload $wl --no-cpu-baseline --no-secondary < /dev/null > $out/prof_$wl.log 2>&1
      db=$(find $out/prof_$wl -name "*.db" | head -1)
      if [ -n "$db" ]; then python tools/rocprof_summary.py $db > $out/${wl}_kernel_trace_stats.txt < /dev/null; tail -1 $out/prof_$wl.log | cut -c1-300; fi
      rm -rf $out/prof_$wl
      echo "trace $wl done: $(left) s left"
    fi
  fi
done
