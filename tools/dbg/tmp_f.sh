cd $GRAFT_REPO_ROOT
for w in cfg5 cfg2; do
for p in 2000 20000 100000; do
for g in 120000 0; do
echo -n "$w pairs=$p gjk_beside_max=$g : "; HFCL_GJK_BESIDE_MAX=$g python bench.py --workload $w --pairs $p --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done; done
python - <<'PY'
# all_primitives (every solid kind, distance): small batches, fan on / off, records byte for byte
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from __graft_entry__ import load_pkg
pkg = load_pkg(); wl, abi = pkg.workloads, pkg.abi
for gen, kind in ((wl.cfg5_mixed_collide, "collide"), (wl.all_primitives, "distance")):
    for n in (3000, 60000):
        b = gen(n=n, seed=3)
        recs = {}
        for g in (120000, 0):
            lib = wl.make_library(pkg, b, options={"gjk_beside_max": g})
            f = lib.collide if b.kind == "collide" else lib.distance
            req = wl.make_request(b, abi)
            recs[g] = f(b.s1, b.s2, b.tf1, b.tf2, req)
            again = f(b.s1, b.s2, b.tf1, b.tf2, req)
            assert again.tobytes() == recs[g].tobytes()
            lib.close()
        print(b.name, n, "fan = in line:", recs[120000].tobytes() == recs[0].tobytes())
PY
