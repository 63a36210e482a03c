#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
run() { env "$@" timeout 100 python bench.py --workload cfg4d --pairs ${N:-100000} --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/j.json 2> $out/j.err; python - "N=${N:-100000} $*" $out/j.json <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    l=json.loads(l[0]); print("%-50s %.3f M q/s  %.2f ms" % (sys.argv[1], l["value"]/1e6, l["ms_per_step"]))
else: print(sys.argv[1], "FAILED", open(sys.argv[2].replace(".json",".err")).read()[-300:])
PY
}
for w in 64 32 16 8 4 2; do run HFCL_BVHD_WINDOW=$w; done
