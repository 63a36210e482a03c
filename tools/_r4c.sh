out=gpurun_out/r4c; mkdir -p $out
export HFCL_BVHD_BUDGET=64
timeout 300 python tools/cfg4d_ids.py 20000 > $out/ids_pool.txt 2>&1
cat $out/ids_pool.txt
for b in 16 64 256; do for lm in 16 24 40; do
  echo "budget $b leaf_min $lm: $(HFCL_BVHD_BUDGET=$b HFCL_BVHD_LEAF_MIN=$lm timeout 300 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"
done; done 2>&1 | tee $out/sweep.txt
echo "coop 1024: $(HFCL_BVHD_POOL=0 HFCL_BVHD_BUDGET=1024 timeout 300 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $out/sweep.txt
