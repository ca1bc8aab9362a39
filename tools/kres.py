#!/usr/bin/env python3
"""Print VGPR / SGPR / spill / LDS figures of the kernels in the built library whose mangled name contains any of the
given fragments (default: the scan kernels).  Usage: python tools/kres.py [fragment ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import re  # noqa: E402
import subprocess  # noqa: E402
import tempfile  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
so = os.path.join(ROOT, "ivfadc.jl_amd", "csrc", "libivfadc_hip.so")
frags = sys.argv[1:] or ["scan_kernel"]
with tempfile.TemporaryDirectory() as tmp:
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
cur = None
rows = []
for line in notes.splitlines():
    if re.match(r"^  - \.\w+:", line):
        cur = {}
        rows.append(cur)
        line = "    " + line[4:]
    if cur is None:
        continue
    m = re.match(r"^    \.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", line)
    if m:
        cur[m.group(1)] = m.group(2)
for r in rows:
    n = r.get("name", "")
    if any(f in n for f in frags):
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        print("%-70s vgpr=%s sgpr=%s spill=%s scratch=%s lds=%s" % (dem[:70], r.get("vgpr_count"), r.get("sgpr_count"),
              r.get("vgpr_spill_count"), r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
