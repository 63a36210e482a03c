#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (*.db) into the plain-text summary committed under profiles/.

usage: rocprof_summary.py <results.db> [<pmc_results.db> ...] > profiles/<name>.txt
  - first table: per-kernel calls / total / average duration (== `rocprofv3 --stats`)
  - then, for every PMC database: per-kernel counter totals and per-dispatch means."""
import collections
import sqlite3
import sys


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    out = ["# kernel-trace stats: %s" % db, "%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct")]
    for name, calls, tot, avg, pct in rows:
        out.append("%-90s %8d %14.1f %12.2f %7.2f" % (name[:90], calls, tot / 1e3, avg / 1e3, pct))
    return "\n".join(out)


def pmc_stats(db):
    con = sqlite3.connect(db)
    rows = list(con.execute(
        "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"))
    d = collections.defaultdict(dict)
    for k, c, v, n in rows:
        d[k][c] = (v, n)
    out = ["# PMC counters: %s" % db]
    for k in sorted(d):
        out.append(k[:110])
        for c in sorted(d[k]):
            v, n = d[k][c]
            out.append("    %-24s total %14.6g   per-dispatch %14.6g   dispatches %d" % (c, v, v / n, n))
    return "\n".join(out)


if __name__ == "__main__":
    dbs = sys.argv[1:]
    for i, db in enumerate(dbs):
        con = sqlite3.connect(db)
        has_pmc = con.execute("select count(*) from counters_collection").fetchone()[0] > 0
        if has_pmc:
            print(pmc_stats(db))
        else:
            print(kernel_stats(db))
        print()
