#!/bin/bash
out=gpurun_out/r5f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
left() { echo "[t=$SECONDS s]"; }
timeout 600 python -m pytest tests/test_bvh_shape.py -q -m gpu -p no:cacheprovider < /dev/null > $out/pytest_bvh_shape.txt 2>&1; tail -12 $out/pytest_bvh_shape.txt | cut -c1-300
left
timeout 500 python tools/mesh_solid_ids.py 100000 box,mixed,convex32 > $out/ids_100k.txt 2>&1; cat $out/ids_100k.txt | cut -c1-420
left
HFCL_SHAPE_DIST_POOL=0 timeout 500 python tools/mesh_solid_ids.py 100000 box > $out/ids_100k_ordered.txt 2>&1; cat $out/ids_100k_ordered.txt | cut -c1-420
left
