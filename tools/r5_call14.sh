#!/bin/bash
out=gpurun_out/r5o
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
timeout 400 python tools/fp64_exactness.py > $out/exact.txt 2>&1; cat $out/exact.txt | cut -c1-300
HFCL_EPA_GENERAL_STAGED_MIN=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp64 or hand_over or split or host_pipeline or edge" -p no:cacheprovider < /dev/null 2>&1 | tail -4 | cut -c1-300
{
for wl in cfg5 cfg2 cfg2f; do
bench staged $wl
bench lockstep $wl HFCL_EPA_GENERAL_STAGED=0
bench staged_again $wl
done
} 2>&1 | tee $out/ab.txt
