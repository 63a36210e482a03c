out=gpurun_out/r4d; mkdir -p $out
run() { timeout 300 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
export HFCL_BVHD_BUDGET=64
timeout 300 python tools/cfg4d_ids.py 20000 > $out/ids_pool.txt 2>&1; cat $out/ids_pool.txt
for lm in 40 48 64; do for sv in 16 32 64; do
  echo "Q4 leaf_min $lm starve $sv: $(HFCL_BVHD_LEAF_MIN=$lm HFCL_BVHD_STARVE=$sv run)"
done; done 2>&1 | tee $out/sweep.txt
for v in q1 q2 q8; do for lm in 24 40; do
  echo "$v leaf_min $lm: $(HFCL_LIB_PATH=build/ab/lib_$v.so HFCL_BVHD_LEAF_MIN=$lm run)"
done; done 2>&1 | tee -a $out/sweep.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
HFCL_BVHD_LEAF_MIN=40 timeout 300 python tools/pmc.py --workload cfg4d --tag r4d SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY > $out/pmc.txt 2>&1
grep -A9 "k_bvh_distance" $out/pmc.txt
HFCL_BVHD_LEAF_MIN=40 timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o t -- python bench.py --workload cfg4d --no-cpu-baseline --no-secondary --steps 5 > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocprof_summary.py $db | head -8; rm -rf $out/prof
