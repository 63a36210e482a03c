timeout 200 python -m pytest tests/test_bvh_shape.py -q -m gpu -p no:cacheprovider < /dev/null 2>&1 | tail -3
for lib in build/ab/lib_old.so ""; do
  echo "== lib '$lib'"
  env ${lib:+HFCL_LIB_PATH=$lib} timeout 100 python bench.py --workload cfg4s --no-cpu-baseline --no-secondary < /dev/null 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4s', d['value'], d['ms_per_step'])"
  env ${lib:+HFCL_LIB_PATH=$lib} timeout 150 python tools/mesh_solid_bench.py --no-distance --kinds mixed,ellipsoid,box,convex32 --reps 5 < /dev/null 2>&1 | grep -v "amdgpu.ids\|^#\|^solid" | cut -c1-60
done
