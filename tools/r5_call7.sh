#!/bin/bash
out=gpurun_out/r5g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
left() { echo "[t=$SECONDS s]"; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider < /dev/null > $out/pytest_gpu.txt 2>&1; tail -25 $out/pytest_gpu.txt | cut -c1-300
left
timeout 500 python tools/mesh_solid_ids.py 100000 box,sphere,capsule > $out/ids_100k.txt 2>&1; cat $out/ids_100k.txt | cut -c1-420
left
HFCL_BVHD_STARVE=0 HFCL_BVHD_LEAF_MIN=64 HFCL_SHAPE_DIST_STARVE=0 HFCL_SHAPE_DIST_LEAF_MIN=64 timeout 120 python -m pytest tests/test_gpu_parity.py tests/test_bvh_shape.py -q -m gpu -k "test_bvh_distance or distance_long_walks" -p no:cacheprovider < /dev/null > $out/pytest_starve0.txt 2>&1; echo "starve0 rc=$?"; tail -3 $out/pytest_starve0.txt
left
