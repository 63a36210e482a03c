# SQ counters of the EPA kernels of a workload (default cfg5): do the waves wait, and for what?   usage: tools/dbg/epa_pmc.sh [workload]
wl=${1:-cfg5}
cd /tmp && export TMPDIR=/tmp
pass() {
  tag=$1; shift
  rm -rf /tmp/ep_$tag
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/ep_$tag -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/ep_$tag.log 2>&1
  db=$(find /tmp/ep_$tag -name "*.db" | head -1)
  if [ -z "$db" ]; then echo "pass $tag: no database"; tail -5 /tmp/ep_$tag.log; return; fi
  python3 - "$db" "$@" <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); names = sys.argv[2:]
rows = con.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by kernel_name, dispatch_id, counter_name").fetchall()
d = collections.OrderedDict()
for k, did, c, v in rows:
    d.setdefault((k.split("(")[0][:48], did), {})[c] = v
agg = collections.OrderedDict()
for (k, did), c in d.items():
    a = agg.setdefault(k, [0, collections.Counter()])
    a[0] += 1
    for n, v in c.items(): a[1][n] += v
print("%-50s %5s " % ("kernel (mean per dispatch, millions)", "n") + " ".join("%20s" % n for n in names))
for k, (cnt, c) in agg.items():
    if max(c.values()) / cnt < 1e5: continue
    print("%-50s %5d " % (k, cnt) + " ".join("%20.3f" % (c[n] / cnt / 1e6) for n in names))
PY
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_IFETCH
pass c SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass d SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT
