#!/usr/bin/env python
"""Compact per-kernel resource table (VGPR / AGPR / scratch / LDS / occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage.
usage: tools/kernel_resources.py [gjk|epa|bvh|bvhd ...] [extra hipcc flags]   (default: all three kernel translation units)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ("gjk32", "gjk64", "epa32", "epa64", "bvh", "bvhc", "bvhs", "bvhd")  # as the Makefile builds them (gjk / epa: both precisions)
ALIAS = {"gjk": ["gjk32", "gjk64"], "epa": ["epa32", "epa64"]}
units = [u for a in sys.argv[1:] for u in ALIAS.get(a, [a] if a in UNITS else [])] or list(UNITS)
flags = [a for a in sys.argv[1:] if a not in UNITS and a not in ALIAS]
procs = []
for u in units:  # the translation units compile side by side
    src = os.path.join(ROOT, "hpp-fcl_amd", "csrc", "hfcl_k_%s.hip" % ("bvh" if u in ("bvhs", "bvhc") else u.rstrip("0123456789")))
    mk = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "hpp-fcl_amd", "csrc"), "-pn"], capture_output=True, text=True).stdout
    m = re.search(r"^FLAGS_k_%s = (.*)$" % u, mk, re.M)  # the Makefile's per-unit flags
    unit_flags = (m.group(1).split() if m else []) + (["-DHFCL_UNIT_PRECISION=" + u[-2:]] if u[-2:] in ("32", "64") else [])
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value", "-Wno-pass-failed",
           "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src] + unit_flags + flags
    procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
err = "".join(p.communicate()[1] for p in procs)
rows, cur = [], {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    else:
        cur[k] = v
print("%-60s %5s %5s %8s %6s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "LDS", "occ"))
for r in rows:
    n = re.sub(r"\(.*", "", r["name"]).replace("void ", "")
    if re.search(r"k_gjk_cvx<\w+, (8|16|32|64),", n):
        continue
    print("%-60s %5s %5s %8s %6s %4s" % (n[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
                                         r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
