cd $GRAFT_REPO_ROOT
for w in cfgmix cfg4s; do
bash tools/dbg/wl_sweep.sh $w "HFCL_MESH_PRIO=1" "HFCL_MESH_PRIO=0" "HFCL_MESH_PRIO=1 HFCL_SHAPE_WALK=0" "HFCL_MESH_PRIO=0 HFCL_SHAPE_WALK=0" "HFCL_MESH_PRIO=1" "HFCL_MESH_PRIO=0"
done
python tools/mesh_solid_bench.py 2>&1 | tail -20
HFCL_SHAPE_WALK=0 python tools/mesh_solid_bench.py 2>&1 | tail -20
