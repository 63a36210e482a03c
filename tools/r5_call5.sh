#!/bin/bash
out=gpurun_out/r5e
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
left() { echo "[t=$SECONDS s]"; }
bench() {
  local label=$1 wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    l=json.loads(sys.stdin.read())
    print('%-6s %-22s %8.1f Mq/s %8.4f ms/step  %s' % ('$wl', '$label', l['value']/1e6, l['ms_per_step'], {k:round(v,3) for k,v in json.load(open('bench_full.json'))['roofline']['kernels_ms'].items() if v > 0.01}))
except Exception as e:
    print('$wl $label FAILED', e)"
}
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider < /dev/null > $out/pytest_gpu.txt 2>&1; tail -30 $out/pytest_gpu.txt | cut -c1-400
left
timeout 300 python tools/fp64_exactness.py > $out/exact_tree.txt 2>&1; cat $out/exact_tree.txt | cut -c1-380
left
timeout 400 python tools/mesh_solid_ids.py 20000 > $out/ids_tree.txt 2>&1; cat $out/ids_tree.txt | cut -c1-420
left
for wl in cfg3 cfg2 cfg5 cfg4 cfg4s cfg4d; do bench tree $wl; done 2>&1 | tee $out/bench.txt
left
