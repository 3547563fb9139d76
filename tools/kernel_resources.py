#!/usr/bin/env python3
"""Registers, scratch and occupancy per kernel of crypto_amd/csrc/bn256msm.hip, from the compiler's own
remarks (hipcc -Rpass-analysis=kernel-resource-usage; gfx950 cross-compile, no GPU needed).

  python tools/kernel_resources.py [substring ...]     only kernels whose mangled name has a substring"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "crypto_amd", "csrc", "bn256msm.hip")
with tempfile.TemporaryDirectory() as tmp:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                        "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "x.so"), src],
                       capture_output=True, text=True)
rows, cur = [], None
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
want = sys.argv[1:]
print("%-58s %6s %6s %8s %5s %7s" % ("kernel", "VGPRs", "AGPRs", "scratch", "occ", "LDS"))
for r in rows:
    if want and not any(w in r["name"] for w in want):
        continue
    print("%-58s %6s %6s %8s %5s %7s" % (r["name"][:58], r.get("VGPRs"), r.get("AGPRs"),
                                          r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
                                          r.get("LDS Size [bytes/block]")))
sys.exit(p.returncode)
