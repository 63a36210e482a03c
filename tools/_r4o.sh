out=gpurun_out/r4o; mkdir -p $out
run() { timeout 200 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
{
echo "== no contraction (the build)"; timeout 200 python tools/cfg4d_ids.py 100000 2>&1 | grep -v amdgpu.ids
echo "cfg4d 100k: $(run)"; echo "cfg4d 1M: $(run '--pairs 1000000')"
echo "lane walk only (HFCL_BVHD_BUDGET=0) 100k: $(HFCL_BVHD_BUDGET=0 run)"
echo "wave-per-walk continuation (HFCL_BVHD_POOL=0, budget 1024) 100k: $(HFCL_BVHD_POOL=0 HFCL_BVHD_BUDGET=1024 run)"
echo "== the same unit built with hipcc's default contraction (-ffp-contract=fast)"
HFCL_LIB_PATH=build/ab/lib_fma.so timeout 200 python tools/cfg4d_ids.py 100000 2>&1 | grep -v amdgpu.ids
echo "cfg4d 100k: $(HFCL_LIB_PATH=build/ab/lib_fma.so run)"; echo "cfg4d 1M: $(HFCL_LIB_PATH=build/ab/lib_fma.so run '--pairs 1000000')"
echo "lane walk only 100k: $(HFCL_LIB_PATH=build/ab/lib_fma.so HFCL_BVHD_BUDGET=0 run)"
echo "wave-per-walk continuation 100k: $(HFCL_LIB_PATH=build/ab/lib_fma.so HFCL_BVHD_POOL=0 HFCL_BVHD_BUDGET=1024 run)"
} 2>&1 | tee $out/contraction.txt
