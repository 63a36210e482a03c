#!/usr/bin/env python
"""Kernel timeline of the LAST step of a bench.py workload from a rocprofv3 kernel-trace CSV: start offset, duration, queue.
usage: rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --workload X --steps 3 --warmup 2 ... ; tools/step_timeline2.py <csv> [first kernel name substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_classify"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in rows[i0:])
print("last step: %.1f us from the start of %s to the last kernel's end" % ((end - t0) / 1e3, first))
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (e - s) < 2500:
        continue
    print("  +%8.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:100]))
