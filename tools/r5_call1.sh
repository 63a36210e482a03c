#!/bin/bash
# round 5, GPU call 1: the staged convex x convex EPA tier (identity with the one-kernel form, A/B, knobs), and the two
# contraction-free experiments (mesh x solid unit; fp64 GJK / EPA units).
out=gpurun_out/r5a
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
timeout 200 python tools/epa_staged_check.py 300000 1 > $out/identity.txt 2>&1; echo "identity rc=$?"; tail -3 $out/identity.txt; left
HFCL_EPA_CC_STAGED_MIN=0 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_epa_ground_truth.py -q -m gpu -k "f32 or fp32 or ground or envelope" -p no:cacheprovider < /dev/null > $out/pytest_staged_min0.txt 2>&1; tail -3 $out/pytest_staged_min0.txt; left
{
bench staged cfg3
bench stream HFCL_EPA_CC_STAGED=0 cfg3
} 2>&1 | tee $out/ab_cfg3.txt
bench staged cfg3 >> $out/ab_cfg3.txt
for v in rm1 rm3 rounds1 rounds3; do bench $v cfg3 HFCL_LIB_PATH=$PWD/build/ab/lib_$v.so | tee -a $out/ab_cfg3.txt; done
bench stream_again cfg3 HFCL_EPA_CC_STAGED=0 | tee -a $out/ab_cfg3.txt
left
# contraction-free fp64 GJK / EPA units
{
for wl in cfg2 cfg5 cfg3; do
  bench tree $wl
  bench ge_nc $wl HFCL_LIB_PATH=$PWD/build/ab/lib_ge_nc.so
done
} 2>&1 | tee $out/ab_ge_nc.txt
timeout 300 python tools/fp64_exactness.py > $out/exact_tree.txt 2>&1; cat $out/exact_tree.txt | cut -c1-330
HFCL_LIB_PATH=$PWD/build/ab/lib_ge_nc.so timeout 300 python tools/fp64_exactness.py > $out/exact_ge_nc.txt 2>&1; cat $out/exact_ge_nc.txt | cut -c1-330
left
# contraction-free mesh unit
{
for wl in cfg4 cfg4s; do
  bench tree $wl
  bench bvh_nc $wl HFCL_LIB_PATH=$PWD/build/ab/lib_bvh_nc.so
done
} 2>&1 | tee $out/ab_bvh_nc.txt
timeout 400 python tools/mesh_solid_ids.py 20000 > $out/ids_tree.txt 2>&1; cat $out/ids_tree.txt | cut -c1-400
HFCL_LIB_PATH=$PWD/build/ab/lib_bvh_nc.so timeout 400 python tools/mesh_solid_ids.py 20000 > $out/ids_bvh_nc.txt 2>&1; cat $out/ids_bvh_nc.txt | cut -c1-400
left
timeout 400 python -m pytest tests -q -m gpu -x -p no:cacheprovider < /dev/null > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
left
