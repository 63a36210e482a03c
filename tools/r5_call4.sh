#!/bin/bash
out=gpurun_out/r5d
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
left() { echo "[t=$SECONDS s]"; }
bench() {  # bench <label> <workload> [env...] [-- bench flags]
  local label=$1 wl=$2; shift 2
  local envs=() flags=()
  while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; flags=("$@"); break; fi; envs+=("$1"); shift; done
  env "${envs[@]}" timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary "${flags[@]}" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    l=json.loads(sys.stdin.read())
    print('%-6s %-22s %8.1f Mq/s %8.4f ms/step  %s' % ('$wl', '$label', l['value']/1e6, l['ms_per_step'], {k:round(v,3) for k,v in json.load(open('bench_full.json'))['roofline']['kernels_ms'].items() if v > 0.01}))
except Exception as e:
    print('$wl $label FAILED', e)"
}
{
bench tree cfg3
bench flat1 cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_flat1.so
bench flat2 cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_flat2.so
bench split2 cfg3 -- --split 2
bench tree_again cfg3
bench split2_again cfg3 -- --split 2
} 2>&1 | tee $out/ab_cfg3.txt
left
