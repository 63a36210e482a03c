#!/bin/bash
out=gpurun_out/${1:-r6soak}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python tools/soak_parity.py 1000000 > $out/soak_parity.txt 2>&1; echo "soak_parity rc=$?"; tail -22 $out/soak_parity.txt | cut -c1-260
timeout 600 python tools/mesh_soak.py --seeds 6 --n 100000 > $out/mesh_soak.txt 2>&1; echo "mesh_soak rc=$?"; tail -8 $out/mesh_soak.txt | cut -c1-200
for s in 21 22 23 24; do timeout 200 python tools/epa_staged_check.py 1000000 $s 2>&1 | tail -1; done | tee $out/staged_identity.txt
echo "[t=$SECONDS s]"
