# kernel timeline of the last cfg4 step (which kernels run beside which)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$1; mkdir -p $out; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $out/tl
env "$@" rocprofv3 --kernel-trace --output-format csv -d $out/tl -- python $root/bench.py --workload cfg4 --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $out/tl.log 2>&1
f=$(find $out/tl -name "*kernel_trace.csv" | head -1)
python $root/tools/step_timeline2.py $f k_classify
rm -rf $out/tl
