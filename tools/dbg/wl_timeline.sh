# kernel timeline of the last step of a bench.py workload (which kernels run beside which):  tools/dbg/wl_timeline.sh <out tag> <workload> [env assignments]
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/$1; wl=$2; shift; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/tl
env A=1 "$@" rocprofv3 --kernel-trace --output-format csv -d $out/tl -- python $root/bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $out/tl_$wl.log 2>&1
f=$(find $out/tl -name "*kernel_trace.csv" | head -1)
python $root/tools/step_timeline2.py $f k_classify
rm -rf $out/tl
