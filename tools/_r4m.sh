out=gpurun_out/r4m; mkdir -p $out
run() { timeout 300 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
export HFCL_BVHD_POOL=2
timeout 120 python tools/cfg4d_ids.py 2000 2>&1 | grep -v amdgpu.ids | tee $out/ids.txt
timeout 300 python tools/cfg4d_ids.py 20000 2>&1 | grep -v amdgpu.ids | tee -a $out/ids.txt
echo "flow: $(run)" | tee $out/sweep.txt
echo "flow 1M: $(timeout 300 python bench.py --workload cfg4d --pairs 1000000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $out/sweep.txt
echo "pool 1M: $(HFCL_BVHD_POOL=1 timeout 300 python bench.py --workload cfg4d --pairs 1000000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $out/sweep.txt
