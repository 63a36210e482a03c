out=gpurun_out/r4l; mkdir -p $out
run() { timeout 300 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for v in q4e1 q4e2 q8e1 q8e2 q8e2r3; do
  HFCL_LIB_PATH=build/ab/lib_$v.so timeout 300 python tools/cfg4d_ids.py 20000 2>&1 | grep "ids equal"
  for pm in 32 48 64; do echo "$v part_min $pm: $(HFCL_BVHD_PART_MIN=$pm HFCL_LIB_PATH=build/ab/lib_$v.so run)"; done
done 2>&1 | tee $out/sweep.txt
for lm in 24 40 56; do for sv in 16 32 64; do echo "q8e2 pm48 leaf_min $lm starve $sv: $(HFCL_BVHD_LEAF_MIN=$lm HFCL_BVHD_STARVE=$sv HFCL_LIB_PATH=build/ab/lib_q8e2.so run)"; done; done 2>&1 | tee -a $out/sweep.txt
