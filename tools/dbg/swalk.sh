cd $GRAFT_REPO_ROOT
bash tools/dbg/wl_sweep.sh cfgmix "HFCL_MESH_OWN_AUX=0" "HFCL_MESH_OWN_AUX=1" "HFCL_MESH_OWN_AUX=0" "HFCL_MESH_OWN_AUX=1" "HFCL_MESH_OWN_AUX=0 HFCL_MESH_BESIDE=4"
bash tools/dbg/wl_timeline.sh mx cfgmix
