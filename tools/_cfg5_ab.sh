#!/bin/bash
out=gpurun_out/$1; shift; mkdir -p $out
run() { timeout 200 python bench.py --workload $1 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $out/j.json 2> $out/j.err; python - "$2 $1" $out/j.json <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    l=json.loads(l[0]); print("%-20s %9.2f M q/s  %.3f ms  %s" % (sys.argv[1], l["value"]/1e6, l["ms_per_step"], {k:round(v,3) for k,v in l["roofline"]["kernels_ms"].items() if v>0.05}))
else: print(sys.argv[1], "FAILED", open(sys.argv[2].replace(".json",".err")).read()[-300:])
PY
}
for v in "$@"; do
  if [ $v = default ]; then unset HFCL_LIB_PATH; else export HFCL_LIB_PATH=$PWD/scratch/lib_$v.so; fi
  run cfg5 $v; run cfg2 $v
done
