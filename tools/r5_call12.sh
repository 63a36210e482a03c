#!/bin/bash
out=gpurun_out/r5l
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
for s in 22 23 25 26; do timeout 200 python tools/epa_staged_check.py 1000000 $s 2>&1 | grep "records identical"; done | tee $out/staged_identity.txt
{
bench resume_walk cfg3
bench resume_par cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_resume_par.so
bench resume_walk_again cfg3
bench resume_par_again cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_resume_par.so
} 2>&1 | tee $out/ab.txt
