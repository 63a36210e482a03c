#!/bin/bash
# On the GPU box: bench every A/B build (tools/ab_build.sh) on the given workloads.
#   tools/ab_run.sh "<lib names, '-' = in-tree build>" "<workloads>" [extra bench.py flags]
libs=$1; wls=$2; shift 2
for wl in $wls; do
  for l in $libs; do
    if [ "$l" = "-" ]; then unset HFCL_LIB_PATH; else export HFCL_LIB_PATH=$PWD/build/ab/libhppfcl_amd_$l.so; fi
    timeout 600 python bench.py --workload $wl --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    l=json.loads(sys.stdin.read())
    print('%-6s %-14s %8.1f Mq/s %8.4f ms/step  %s' % ('$wl', '$l', l['value']/1e6, l['ms_per_step'], {k:round(v,3) for k,v in l['roofline']['kernels_ms'].items() if v > 0.02}))
except Exception as e:
    print('$wl $l FAILED', e)"
  done
done
