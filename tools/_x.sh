cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { python bench.py --workload $1 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | cut -c1-200; }
for rep in 1 2 3; do
for v in before base; do
  if [ $v = base ]; then unset HFCL_LIB_PATH; else export HFCL_LIB_PATH=$PWD/build/ab/lib_$v.so; fi
  echo "== $v cfg3: $(run cfg3)   cfg2f: $(run cfg2f)"
done
done
unset HFCL_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_epa_ground_truth.py -q -m gpu 2>&1 | tail -3
for s in 21 22; do timeout 200 python tools/epa_staged_check.py 1000000 $s 2>&1 | tail -1; done
