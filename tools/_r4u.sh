out=gpurun_out/r4u; mkdir -p $out
run() { timeout 200 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
{ for v in q8e1 q2e1; do echo "$v 100k: $(HFCL_LIB_PATH=build/ab/lib_$v.so run)"; done
for b in 16 32 128; do echo "budget $b: $(HFCL_BVHD_BUDGET=$b run)"; done
for lm in 16 32 48; do for sv in 16 32 48; do echo "leaf_min $lm starve $sv: $(HFCL_BVHD_LEAF_MIN=$lm HFCL_BVHD_STARVE=$sv run)"; done; done
for pm in 0 32 64; do echo "part_min $pm: $(HFCL_BVHD_PART_MIN=$pm run)"; done
} 2>&1 | tee $out/sweep.txt
