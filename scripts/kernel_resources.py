#!/usr/bin/env python3
"""Register / LDS / spill figures of every gfx950 kernel in an object file (from the code object's metadata notes).
    python scripts/kernel_resources.py zeldovich_plt_amd/csrc/build/zd_kernels.o [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ZD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def main(obj, pats):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    cur = {}
    rows = []
    for line in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            rows.append(cur)
            cur = {}
        cur[k] = v
    if cur.get("name"):
        rows.append(cur)
    for r in rows:
        name = subprocess.run(["c++filt", r.get("name", "?").strip("'\"")], capture_output=True, text=True).stdout.strip()
        if pats and not any(p in name for p in pats):
            continue
        print("%-70s vgpr %4s agpr %3s sgpr %4s spill %3s lds %6s scratch %5s wg %5s" % (
            name[:70], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"), r.get("vgpr_spill_count"),
            r.get("group_segment_fixed_size"), r.get("private_segment_fixed_size"), r.get("max_flat_workgroup_size")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
