#!/bin/bash
# On the GPU box: what a phase of the EPA trip costs, by running it twice per iteration (build/ab/lib_ph{1,2,4}.so: support, horizon search,
# closest-face scan; tools/build_variant.sh phN k_epa32 -DHFCL_EPA_PHASE_TWICE=N).  PC sampling is not available on this box.
out=gpurun_out/${1:-phases}
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
bench base cfg3
bench support_twice cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_ph1.so
bench horizon_twice cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_ph2.so
bench closest_twice cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_ph4.so
bench base_again cfg3
} 2>&1 | tee $out/phases.txt
