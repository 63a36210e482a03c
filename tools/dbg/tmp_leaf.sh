cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -2
bash tools/dbg/wl_sweep.sh cfg4 "A=1" "A=1" "A=1"
bash tools/dbg/wl_sweep.sh cfgmix "A=1" "A=1"
timeout 600 python tools/mesh_soak.py --seeds 3 --n 100000 2>&1 | tail -5
