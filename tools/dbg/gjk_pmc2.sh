# SQ counters of k_gjk_cvx per iteration cap, several passes: what do the waves wait for from the first tetrahedron trip on?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ[C]*_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/avail_sq.txt
pass() {
  tag=$1; shift
  rm -rf /tmp/gp_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/gp_$tag -o t -- python $GRAFT_REPO_ROOT/tools/dbg/gjk_setup_cost.py > /tmp/gp_$tag.log 2>&1
  db=$(find /tmp/gp_$tag -name "*.db" | head -1)
  if [ -z "$db" ]; then echo "pass $tag: no database"; grep -i "error\|invalid\|not" /tmp/gp_$tag.log | head -5; return; fi
  python3 - "$db" "$@" <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); names = sys.argv[2:]
rows = con.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by kernel_name, dispatch_id, counter_name").fetchall()
d = collections.OrderedDict()
for k, did, c, v in rows:
    if "k_gjk_cvx" in k:
        d.setdefault(did, {})[c] = v
ids = sorted(d)
big = [i for i in ids if max(d[i].values()) > 1e5]
caps = (1, 2, 3, 4, 6, 8, 12, 128)
print("cap " + " ".join("%22s" % n for n in names))
for i, cap in enumerate(caps):
    if 4 * i + 3 >= len(big): break
    c = d[big[4 * i + 3]]
    print("%3d " % cap + " ".join("%22.3f" % (c.get(n, float('nan')) / 1e6) for n in names))
PY
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
pass b SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH
pass c SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
pass d SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
pass e SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
