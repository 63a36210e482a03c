run() { echo -n "$* : "; n=$1; shift; env "$@" python bench.py --workload cfg4 --pairs $n --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value']/1e6)"; }
for n in 20000 100000 250000 1000000; do
run $n A=1
run $n HFCL_BVH_WALK_ROUNDS=0
done
run 1000000 HFCL_BVH_BUDGET0_COOP=320
run 1000000 HFCL_BVH_BUDGET0_COOP=448 HFCL_BVH_WALK_BUDGET=512
run 1000000 HFCL_BVH_BUDGET0_COOP=640 HFCL_BVH_WALK_BUDGET=512
run 250000 HFCL_BVH_BUDGET0_COOP=320
run 250000 HFCL_BVH_BUDGET0_COOP=320 HFCL_BVH_WALK_BUDGET=384
