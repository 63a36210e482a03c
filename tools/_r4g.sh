out=gpurun_out/r4g; mkdir -p $out
run() { timeout 300 python bench.py --workload $1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
timeout 300 python tools/cfg4d_ids.py 20000 2>&1 | grep -v amdgpu.ids | tee $out/ids.txt
echo "cfg4d: $(run cfg4d)" | tee $out/sweep.txt
echo "cfg4d coop1024: $(HFCL_BVHD_POOL=0 HFCL_BVHD_BUDGET=1024 run cfg4d)" | tee -a $out/sweep.txt
echo "cfg4s: $(run cfg4s)" | tee -a $out/sweep.txt
