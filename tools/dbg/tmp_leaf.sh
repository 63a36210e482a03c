cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -k "mixed or shape or solid or mesh" 2>&1 | tail -2
bash tools/dbg/wl_sweep.sh cfg4s "A=1" "A=1" "A=1"
bash tools/dbg/wl_sweep.sh cfgmix "A=1" "A=1"
timeout 600 python tools/mesh_soak.py --seeds 3 --n 100000 2>&1 | tail -5
bash tools/dbg/wl_timeline.sh lf cfg4s | grep "leaves\|last step"
