#!/usr/bin/env python
"""Registers, scratch and LDS of every kernel instantiation, and the residency both allow (no GPU needed: the figures come
from the compiler's -Rpass-analysis=kernel-resource-usage remarks of the kernel units, built with their Makefile flags).

  tools/occupancy_survey.py            compile and print the table (profiles/r03_h_occupancy_survey.txt)

waves/SIMD by registers = 512 / (VGPRs + AGPRs); waves/CU by LDS = (128 / LDS allocation units of 1280 B per block) x waves per
block (tools/occupancy_probe.hip).  A kernel whose two limits disagree wastes the smaller resource's headroom: this survey is
how round 3 found the mesh distance kernel at 3 waves per CU (48 KB of LDS per wave) and k_gjk_large<double> at one wave per SIMD
(260 registers) -- 3.3x and 1.75x once fixed."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hpp-fcl_amd", "csrc")
FLAGS = {"k_gjk": "-fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt",
         "k_epa": "-fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=on", "k_bvh": "", "k_util": ""}
BLOCK = {"k_bvh_collide": 128, "k_bvh_distance": 64, "k_bvh_shape": 64, "k_bvh_shape_distance": 64, "k_triangle": 64, "k_epa": 64,
         "k_epa_stream": 64, "k_classify": 1024, "k_bvh_coop": 64, "k_bvh_shape_coop": 64, "k_bvh_distance_coop": 64,
         "k_bvh_shape_distance_coop": 64, "k_bvh_shape_distance_lane": 64, "k_bvh_shape_finish": 64, "k_bvh_level_mark": 64}


def main():
    rows = []
    for unit, fl in FLAGS.items():
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value", "-Wno-pass-failed",
               "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", "hfcl_%s.hip" % unit] + fl.split()
        txt = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
        for blk in txt.split("Function Name: ")[1:]:
            name = blk.split()[0]
            g = lambda k: int(re.search(k + r": (\d+)", blk).group(1)) if re.search(k + r": (\d+)", blk) else 0  # noqa: E731
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"^void ", "", re.sub(r"\((Work|hfcl_result|double const).*", "", dem))
            rows.append((dem, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]")))
    print("%-52s %5s %5s %8s %8s   %s" % ("kernel", "VGPR", "AGPR", "scratch", "LDS/blk", "waves/SIMD by registers | waves/CU by LDS"))
    for dem, vg, ag, sc, lds in rows:
        base = re.match(r"(\w+)", dem).group(1)
        wpb = BLOCK.get(base, 256) // 64
        wreg = min(512 // max(vg + ag, 1), 8)
        units = (lds + 1279) // 1280
        wlds = (128 // units) * wpb if units else None
        flag = "  <-- LDS below registers" if wlds is not None and wlds < 4 * wreg else ""
        print("%-52s %5d %5d %8d %8d   %d (%d per CU) | %s%s" % (dem[:52], vg, ag, sc, lds, wreg, 4 * wreg, wlds if wlds is not None else "-", flag))


if __name__ == "__main__":
    sys.exit(main())
