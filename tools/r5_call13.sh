#!/bin/bash
out=gpurun_out/r5n
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
HFCL_LIB_PATH=$PWD/build/ab/lib_hzvis.so timeout 200 python tools/epa_staged_check.py 300000 1 2>&1 | grep "records identical"
HFCL_LIB_PATH=$PWD/build/ab/lib_hzvis.so HFCL_EPA_CC_STAGED_MIN=0 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_epa_ground_truth.py -q -m gpu -k "f32 or fp32 or ground or envelope" -p no:cacheprovider < /dev/null 2>&1 | tail -2
{
bench base cfg3
bench hzvis cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_hzvis.so
bench base_again cfg3
bench hzvis_again cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_hzvis.so
} 2>&1 | tee $out/ab.txt
