P='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline"]["kernels_ms"]; print("%.1fM q/s"%(d["value"]/1e6), {a:round(b,3) for a,b in k.items() if b>0.02})'
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | head -30) | tee gpurun_out/pytest_gpu.log
for w in cfg3 cfg2 cfg5; do echo "== $w"; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $w 2>/dev/null | python -c "$P"; done
