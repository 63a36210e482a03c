# SQ counters of k_gjk_cvx per iteration cap (tools/dbg/gjk_setup_cost.py under rocprofv3 --pmc): is a trip level instructions or waiting?
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/gp -o t -- python $GRAFT_REPO_ROOT/tools/dbg/gjk_setup_cost.py > /tmp/gp.log 2>&1
grep -v "^[EWI]2026" /tmp/gp.log | tail -8
db=$(find /tmp/gp -name "*.db" | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by kernel_name, dispatch_id, counter_name").fetchall()
d = collections.OrderedDict()
for k, did, c, v in rows:
    if "k_gjk_cvx" in k:
        d.setdefault(did, {})[c] = v
ids = [i for i in sorted(d) if d[i].get("SQ_INSTS_VALU", 0) > 5e6]  # (the batch's other k_gjk_cvx launch has no pair)
caps = (1, 2, 3, 4, 6, 8, 12, 128)
print(len(ids), "dispatches")
print("cap   VALU M   SALU M   LDS M   wave-cycles G   busy-cycles M")
for i, cap in enumerate(caps):
    c = d[ids[4 * i + 3]]
    print("%3d  %7.1f  %7.1f  %6.1f  %10.3f  %10.1f" % (cap, c["SQ_INSTS_VALU"] / 1e6, c["SQ_INSTS_SALU"] / 1e6, c["SQ_INSTS_LDS"] / 1e6, c["SQ_WAVE_CYCLES"] / 1e9, c["SQ_BUSY_CYCLES"] / 1e6))
PY
