out=gpurun_out/r4k; mkdir -p $out
run() { timeout 300 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for v in prof_q8e2c prof_q8e1c prof_q16e1c; do echo "== $v"; HFCL_LIB_PATH=build/ab/lib_$v.so python tools/pool_prof.py 100000 2>&1 | grep -v amdgpu.ids;  echo "$v: $(HFCL_LIB_PATH=build/ab/lib_$v.so run)"; done | tee $out/prof.txt
