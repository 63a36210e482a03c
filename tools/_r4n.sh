out=gpurun_out/r4n; mkdir -p $out
run() { timeout 100 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
export HFCL_BVHD_POOL=2
timeout 100 python tools/cfg4d_ids.py 20000 2>&1 | grep -v amdgpu.ids | tee $out/ids.txt
for lm in 1 8 16 24 32 48; do echo "flow leaf_min $lm: $(HFCL_BVHD_LEAF_MIN=$lm run)"; done | tee $out/sweep.txt
