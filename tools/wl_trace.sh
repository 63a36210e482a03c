#!/bin/bash
# On the GPU box: per-kernel durations (rocprofv3 --kernel-trace) of bench.py workloads, median over the timed steps.
#   tools/wl_trace.sh <out tag> <workload>[:pairs] ...        e.g.  tools/wl_trace.sh r06_a cfg4d cfg4s cfg4:250000
out=gpurun_out/$1; mkdir -p $out; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  wl=${spec%%:*}; n=${spec#*:}; [ "$n" = "$spec" ] && n=0
  rm -rf $root/$out/tr_$wl
  rocprofv3 --kernel-trace --output-format csv -d $root/$out/tr_$wl -- python $root/bench.py --workload $wl --pairs $n --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $root/$out/tr_$wl.log 2>&1
  f=$(find $root/$out/tr_$wl -name "*kernel_trace.csv" | head -1)
  echo "== $spec: $(tail -1 $root/$out/tr_$wl.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("%.4f ms/step  %.3g q/s" % (d["ms_per_step"], d["value"]))' 2>/dev/null)"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:84]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = {k: sorted(v)[len(v) // 2] for k, v in d.items()}
for k, v in sorted(d.items(), key=lambda kv: -tot[kv[0]]):
    if tot[k] >= 3.0:
        print("  %-86s n=%3d  median %9.1f us  min %9.1f" % (k, len(v), tot[k], min(v)))
PY
  rm -rf $root/$out/tr_$wl
done
