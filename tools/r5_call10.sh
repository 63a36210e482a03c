#!/bin/bash
out=gpurun_out/r5j
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
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
{
bench tree cfg4s
bench prev cfg4s HFCL_LIB_PATH=$PWD/build/ab/lib_prev.so
bench tree_again cfg4s
bench tree_nocut cfg4s HFCL_SHAPE_CUT_TICKS=0
} 2>&1 | tee $out/ab.txt
timeout 300 python tools/cut_check.py > $out/cut_check.txt 2>&1; tail -5 $out/cut_check.txt | cut -c1-300
