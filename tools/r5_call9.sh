#!/bin/bash
out=gpurun_out/r5i
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
timeout 200 python tools/epa_staged_check.py 300000 1 > $out/identity.txt 2>&1; echo "identity rc=$?"; tail -1 $out/identity.txt; left
{
bench tree cfg3
bench prev cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_prev.so
bench noatomic_TIMING_ONLY cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_noatomic.so
bench tree_again cfg3
} 2>&1 | tee $out/ab.txt
left
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider < /dev/null > $out/pytest_gpu.txt 2>&1; tail -6 $out/pytest_gpu.txt | cut -c1-300
left
