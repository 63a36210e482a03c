cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
out=gpurun_out/r4b; mkdir -p $out
for wl in cfg4d cfg4s; do
  timeout 600 python tools/measure_traffic.py --workload $wl --out $out/traffic_$wl.json > $out/traffic_$wl.log 2>&1 || echo "traffic $wl FAILED"
done
timeout 300 python tools/pmc.py --workload cfg4d --tag r4b SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU > $out/pmc_cfg4d.txt 2>&1
for wl in cfg4d cfg4s; do
  rm -rf $out/prof_$wl
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$wl -o $wl -- python bench.py --workload $wl --no-cpu-baseline --no-secondary > $out/prof_$wl.log 2>&1
  db=$(find $out/prof_$wl -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > $out/${wl}_kernel_trace_stats.txt
  rm -rf $out/prof_$wl
done
cat $out/cfg4d_kernel_trace_stats.txt $out/cfg4s_kernel_trace_stats.txt
