out=gpurun_out/r4h; mkdir -p $out
HFCL_LIB_PATH=build/ab/lib_prof.so python tools/pool_prof.py 100000 2>&1 | grep -v amdgpu.ids | tee $out/prof.txt
HFCL_BVHD_LEAF_MIN=48 HFCL_LIB_PATH=build/ab/lib_prof.so python tools/pool_prof.py 100000 2>&1 | grep -v amdgpu.ids | tee -a $out/prof.txt
