cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
python tools/dbg/cfg3_checksum.py 2>&1 | grep cfg3
for w in cfg3 cfg5 cfg3u cfg2f; do bash tools/dbg/wl_sweep.sh $w "A=1" "A=1"; done
