#!/bin/bash
# A/B of the mesh x mesh collide kernel on the GPU box: HFCL_BVH_FILTER=0 (plain fp64 tests) against the fp32 filter,
# at the BASELINE size (100k queries) and at 1M queries.  Usage: tools/ab_cfg4.sh <out-dir> [extra env assignments...]
out=$1; shift
mkdir -p $out
for f in 0 1; do
  for n in 100000 1000000; do
    env HFCL_BVH_FILTER=$f "$@" python bench.py --workload cfg4 --pairs $n --steps 10 --warmup 2 --no-cpu-baseline > $out/cfg4_f${f}_n${n}.json 2> $out/cfg4_f${f}_n${n}.err
    python - <<PY
import json
l=[x for x in open("$out/cfg4_f${f}_n${n}.json") if x.startswith("{")]
if l:
    l=json.loads(l[0]); print("filter=$f n=$n  %.1f M q/s  %.3f ms/step  kernels %s" % (l["value"]/1e6, l["ms_per_step"], {k:round(v,3) for k,v in l["roofline"]["kernels_ms"].items()}))
else:
    print("filter=$f n=$n FAILED", open("$out/cfg4_f${f}_n${n}.err").read()[-800:])
PY
  done
done
