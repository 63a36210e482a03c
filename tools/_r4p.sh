out=gpurun_out/r4p; mkdir -p $out
run() { timeout 200 python bench.py --workload cfg4d --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --pairs 1000000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
{ echo "Q4E1 1M: $(run)"; for v in q8e1 q8e2 q8e2r1; do for pm in 32 64; do echo "$v part_min $pm 1M: $(HFCL_BVHD_PART_MIN=$pm HFCL_LIB_PATH=build/ab/lib_$v.so run)"; done; done; } 2>&1 | tee $out/sweep1M.txt
