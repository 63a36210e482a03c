#!/bin/bash
out=gpurun_out/r5c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
left() { echo "[t=$SECONDS s]"; }
bench() {  # bench <label> <workload> [env...]
  local label=$1 wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    l=json.loads(sys.stdin.read())
    print('%-6s %-22s %8.1f Mq/s %8.4f ms/step  %s' % ('$wl', '$label', l['value']/1e6, l['ms_per_step'], {k:round(v,3) for k,v in json.load(open('bench_full.json'))['roofline']['kernels_ms'].items() if v > 0.01}))
except Exception as e:
    print('$wl $label FAILED', e)"
}
timeout 200 python tools/epa_staged_check.py 300000 1 > $out/identity.txt 2>&1; echo "identity rc=$?"; tail -4 $out/identity.txt | cut -c1-600; left
timeout 200 python tools/epa_staged_check.py 1000000 3 > $out/identity1M.txt 2>&1; echo "identity 1M rc=$?"; tail -2 $out/identity1M.txt | cut -c1-300; left
HFCL_EPA_CC_STAGED_MIN=0 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_epa_ground_truth.py -q -m gpu -k "f32 or fp32 or ground or envelope" -p no:cacheprovider < /dev/null > $out/pytest_staged_min0.txt 2>&1; tail -3 $out/pytest_staged_min0.txt; left
{
bench tree cfg3
bench noflat cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_noflat.so
bench tree_inline cfg3 HFCL_EPA_RECORDS_ASIDE=0
bench tree_again cfg3
bench noflat_again cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_noflat.so
bench stream_flat cfg3 HFCL_EPA_CC_STAGED=0
bench stream_noflat cfg3 HFCL_EPA_CC_STAGED=0 HFCL_LIB_PATH=$PWD/build/ab/lib_noflat.so
} 2>&1 | tee $out/ab_cfg3.txt
left
timeout 400 python -m pytest tests -q -m gpu -x -p no:cacheprovider < /dev/null > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
left
