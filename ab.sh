P='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline"]["kernels_ms"]; print("%.1fM q/s"%(d["value"]/1e6), {a:round(b,3) for a,b in k.items() if "cvx" in a})'
echo "== default cfg5"; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload cfg5 2>/dev/null | python -c "$P"
for v in g64_1 g64_2 g64_4; do
echo "== $v cfg5"; HFCL_LIB_PATH=$PWD/gpurun_in_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload cfg5 2>/dev/null | python -c "$P"
HFCL_LIB_PATH=$PWD/gpurun_in_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fp64" 2>&1 | tail -1
done
