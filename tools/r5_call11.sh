#!/bin/bash
out=gpurun_out/r5k
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
bench two_streams cfg5
bench one_stream cfg5 HFCL_EPA64_TWO_STREAMS=0
bench two_streams_again cfg5
bench one_stream_again cfg5 HFCL_EPA64_TWO_STREAMS=0
bench two_streams_nosplit cfg5 HFCL_SPLIT=1
bench one_stream_nosplit cfg5 HFCL_SPLIT=1 HFCL_EPA64_TWO_STREAMS=0
} 2>&1 | tee $out/ab.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp64 or split or hand_over" -p no:cacheprovider < /dev/null 2>&1 | tail -3
