# tools/dbg/wl_sweep.sh <workload> "<env assignments>" ...  -- ms per step of a bench.py workload under each set of option overrides (environment fallback)
wl=$1; shift
for e in "$@"; do echo -n "$wl [$e] : "; env A=1 $e python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done
