#!/usr/bin/env python
"""The kernels of the LAST step of a rocprofv3 --kernel-trace run (rocpd database), in launch order: start offset and duration of each.
usage: tools/step_timeline.py <results.db> [first-kernel-of-a-step substring, default k_classify]"""
import sqlite3
import sys

db = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_classify"
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
idx = [i for i, r in enumerate(rows) if first in r[0]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][1]
for n, st, en, q in rows[i0:i1]:
    print("%-60s q%-2d start %9.1f us  dur %9.1f us" % (n[:60], q, (st - t0) / 1e3, (en - st) / 1e3))
print("step: %.1f us from the first kernel's start to the last kernel's end" % ((max(r[2] for r in rows[i0:i1]) - t0) / 1e3))
