cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | cut -c1-400
for w in cfg3 cfg2 cfg4 cfg5; do timeout 600 python tools/measure_traffic.py --workload $w > gpurun_out/traffic_$w.log 2>&1 || tail -3 gpurun_out/traffic_$w.log; cp gpurun_out/traffic_$w.json profiles/traffic_$w.json; done
for w in cfg3 cfg2 cfg4 cfg5; do timeout 900 python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; tail -c 300 gpurun_out/bench_$w.json; echo; done
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o cfg3 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_final -name "*.db") > gpurun_out/r01_g_cfg3_stats.txt; head -8 gpurun_out/r01_g_cfg3_stats.txt | cut -c1-150
