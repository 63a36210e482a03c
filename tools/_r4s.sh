out=gpurun_out/r4s; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 python tools/pmc.py --workload cfg4s --tag r4s SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD > $out/pmc_cfg4s.txt 2>&1
grep -A9 "k_bvh_shape_coop\|k_bvh_shape_finish\|k_bvh_collide<double, false, false, true" $out/pmc_cfg4s.txt
timeout 300 python tools/pmc.py --workload cfg4 --tag r4s SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES > $out/pmc_cfg4.txt 2>&1
grep -A5 "k_bvh_coop\|k_bvh_collide<double, false, false, false" $out/pmc_cfg4.txt
for b in 4 16 64; do echo "cfg4s shape budget0 $b: $(HFCL_SHAPE_BUDGET0=$b timeout 100 python bench.py --workload cfg4s --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])")"; done
