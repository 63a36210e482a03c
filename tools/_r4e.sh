out=gpurun_out/r4e; mkdir -p $out
run() { timeout 300 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "default: $(run)" | tee $out/sweep.txt
echo "w3: $(HFCL_LIB_PATH=build/ab/lib_w3.so run)" | tee -a $out/sweep.txt
echo "1M default: $(timeout 300 python bench.py --workload cfg4d --pairs 1000000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-300)" | tee -a $out/sweep.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $out/pytest.txt
