cd $GRAFT_REPO_ROOT
L0=$GRAFT_REPO_ROOT/build/ab/lib_lsb0.so
bash tools/dbg/wl_sweep.sh cfg4s "A=1" "HFCL_LIB_PATH=$L0" "A=1" "HFCL_LIB_PATH=$L0"
bash tools/dbg/wl_sweep.sh cfgmix "A=1" "HFCL_LIB_PATH=$L0"
python tools/mesh_solid_bench.py --kinds convex32,box 2>&1 | grep "collide\|distance"
HFCL_LIB_PATH=$L0 python tools/mesh_solid_bench.py --kinds convex32,box 2>&1 | grep "collide\|distance"
timeout 900 python -m pytest tests -m gpu -q -x -k "mixed or shape or solid or mesh" 2>&1 | tail -2
