run() { timeout 200 python bench.py --workload $1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "cfg4 100k: $(run cfg4)"; echo "cfg4 250k: $(run cfg4 '--pairs 250000')"; echo "cfg4 10k: $(run cfg4 '--pairs 10000')"
echo "cfg4d coop form 100k: $(HFCL_BVHD_POOL=0 HFCL_BVHD_BUDGET=1024 run cfg4d)"
echo "cfg4s 100k: $(run cfg4s)"
python tools/mesh_solid_bench.py 2>&1 | grep -v amdgpu | tail -14
