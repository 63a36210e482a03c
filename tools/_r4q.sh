out=gpurun_out/r4q; mkdir -p $out
run() { timeout 200 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
{ timeout 200 python tools/cfg4d_ids.py 20000 2>&1 | grep "ids equal"
echo "rolled triangle loop 100k: $(run)"; echo "rolled 1M: $(run '--pairs 1000000')"
echo "unrolled (HEAD) 100k: $(HFCL_LIB_PATH=build/ab/lib_unrolled.so run)"; echo "unrolled 1M: $(HFCL_LIB_PATH=build/ab/lib_unrolled.so run '--pairs 1000000')"
echo "rolled, coop1024: $(HFCL_BVHD_POOL=0 HFCL_BVHD_BUDGET=1024 run)"; echo "unrolled, coop1024: $(HFCL_LIB_PATH=build/ab/lib_unrolled.so HFCL_BVHD_POOL=0 HFCL_BVHD_BUDGET=1024 run)"
} 2>&1 | tee $out/rolled.txt
