#!/usr/bin/env python3
"""Which kernels sit a few registers above a boundary that would admit another workgroup per CU?
   python scripts/occupancy_scan.py zeldovich_plt_amd/csrc/build/zd_kernels.o
(512 VGPRs per SIMD lane, allocation granule 8, 4 SIMDs per CU; LDS is dynamic here and not considered: check the launcher.)"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main(obj, slack=12):
    out = subprocess.check_output([sys.executable, os.path.join(HERE, "kernel_resources.py"), obj], text=True)
    for l in out.split("\n"):
        m = re.match(r"(.*?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)\s+sgpr\s+(\d+)\s+spill\s+(\d+).*wg\s+(\d+)", l)
        if not m:
            continue
        name, v, a, wg = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(6))
        if wg < 128:
            continue
        waves = (wg + 63) // 64
        alloc = ((v + a + 7) // 8) * 8
        wgs = (min(512 // alloc, 8) * 4) // waves
        for cap in (64, 80, 96, 128, 168):
            if cap < v <= cap + slack:
                new = ((512 // cap) * 4) // waves
                if new > wgs:
                    print("%-60s vgpr %3d threads %4d: %d workgroup(s) per CU, %d if held to %d registers" % (name[:60], v, wg, wgs, new, cap))


if __name__ == "__main__":
    main(sys.argv[1])
