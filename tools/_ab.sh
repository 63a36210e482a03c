# usage: tools/_ab.sh <lib> <label> [env...]
lib=$1; label=$2; shift; shift
for n in 100000 1000000; do
env HFCL_LIB_PATH=$lib "$@" python bench.py --workload cfg4 --pairs $n --steps 10 --warmup 2 --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err
python - <<PY
import json
l=[x for x in open("/tmp/o.json") if x.startswith("{")]
if l:
    l=json.loads(l[0]); print("$label n=$n  %.1f M q/s  %.3f ms/step  k_bvh_collide %.3f" % (l["value"]/1e6, l["ms_per_step"], l["roofline"]["kernels_ms"]["k_bvh_collide"]))
else: print("$label n=$n FAILED", open("/tmp/o.err").read()[-600:])
PY
done
