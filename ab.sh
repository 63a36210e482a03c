mkdir -p gpurun_out
P='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline"]["kernels_ms"]; print("%.1fM q/s"%(d["value"]/1e6), {a:round(b,3) for a,b in k.items() if b>0.02}, d["config"]["buckets"].get("epa_overflow"))'
echo "== default (cap20, W=4, WE=8)"; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
for v in we4 we2 we16 we4c16; do echo "== $v"; HFCL_LIB_PATH=$PWD/gpurun_in_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
