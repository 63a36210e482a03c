run() { timeout 200 python bench.py --workload cfg4d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
timeout 200 python tools/cfg4d_ids.py 20000 2>&1 | grep "ids equal"
echo "100k: $(run)"; echo "1M: $(run '--pairs 1000000')"; echo "10k: $(run '--pairs 10000')"
