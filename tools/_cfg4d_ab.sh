#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
run() { env "$@" python bench.py --workload cfg4d --pairs ${N:-100000} --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/j.json 2> $out/j.err; python - "N=${N:-100000} $*" $out/j.json <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    l=json.loads(l[0]); print("%-50s %.3f M q/s  %.2f ms  %s" % (sys.argv[1], l["value"]/1e6, l["ms_per_step"], {k:round(v,1) for k,v in l["roofline"]["kernels_ms"].items() if v>1}))
else: print(sys.argv[1], "FAILED", open(sys.argv[2].replace(".json",".err")).read()[-400:])
PY
}
for b in 32 64 128 256 768 1024; do run HFCL_BVHD_BUDGET=$b; done
N=10000 run HFCL_BVHD_BUDGET=0
N=10000 run HFCL_BVHD_BUDGET=256
N=10000 run HFCL_BVHD_BUDGET=1024
