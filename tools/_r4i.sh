out=gpurun_out/r4i; mkdir -p $out
run() { timeout 300 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
timeout 300 python tools/cfg4d_ids.py 20000 2>&1 | grep -v amdgpu.ids | tee $out/ids.txt
echo "default (Q4 E2): $(run)" | tee $out/sweep.txt
for v in e1 e3 q2e2 q8e2; do echo "$v: $(HFCL_LIB_PATH=build/ab/lib_$v.so run)" | tee -a $out/sweep.txt; done
for lm in 16 32 48 64; do for sv in 32 64 96; do echo "Q4E2 leaf_min $lm starve $sv: $(HFCL_BVHD_LEAF_MIN=$lm HFCL_BVHD_STARVE=$sv run)" | tee -a $out/sweep.txt; done; done
HFCL_LIB_PATH=build/ab/lib_prof.so python tools/pool_prof.py 100000 2>&1 | grep -v amdgpu.ids | tee $out/prof.txt
