out=gpurun_out/r4j; mkdir -p $out
for v in prof_e1 prof_e1c160 prof_q8e1 prof_q8e2; do echo "== $v"; HFCL_LIB_PATH=build/ab/lib_$v.so python tools/pool_prof.py 100000 2>&1 | grep -v amdgpu.ids; done | tee $out/prof.txt
