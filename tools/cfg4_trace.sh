#!/bin/bash
out=gpurun_out/$1; mkdir -p $out; shift
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/tr_$n -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4 --pairs $n --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$out/tr_$n.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$out/tr_$n -name "*kernel_trace.csv" | head -1)
  echo "== n=$n"
  python - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    if "k_bvh" in r["Kernel_Name"]: d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print("%-62s n=%3d  median %.1f us  min %.1f" % (k, len(v), sorted(v)[len(v)//2], min(v)))
PY
  rm -rf $GRAFT_REPO_ROOT/$out/tr_$n
done
