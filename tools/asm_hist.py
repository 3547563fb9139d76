#!/usr/bin/env python3
"""Opcode histogram per basic block of one kernel, from the gfx950 assembly hipcc leaves with --save-temps
(cross-compiles here, no GPU needed): how many VALU / SALU / memory instructions a loop body holds.

  python tools/asm_hist.py <kernel-name-substring> [min-block-size] [--s file.s]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:]]
sfile = None
if "--s" in args:
    i = args.index("--s")
    sfile = args[i + 1]
    del args[i:i + 2]
want = args[0]
minsize = int(args[1]) if len(args) > 1 else 40
if sfile is None:
    tmp = tempfile.mkdtemp()
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "--save-temps",
                    "-o", "x.so", os.path.join(ROOT, "crypto_amd", "csrc", "bn256msm.hip")], cwd=tmp, check=True,
                   capture_output=True)
    sfile = os.path.join(tmp, "bn256msm-hip-amdgcn-amd-amdhsa-gfx950.s")
    print("assembly:", sfile)
s = open(sfile).read().split("\n")
starts = [i for i, l in enumerate(s) if re.match(r"^_Z\w*:", l) and want in l.split(":")[0]]
for st in starts:
    en = next(i for i in range(st, len(s)) if s[i].strip().startswith(".Lfunc_end"))
    print("==", s[st].split(":")[0], en - st, "lines")
    blocks, cur = [], ("entry", [])
    for l in s[st + 1:en]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if re.match(r"^\.?[A-Za-z_0-9$]+:", t):
            blocks.append(cur)
            cur = (t.split(":")[0], [])
            continue
        if t.startswith("."):
            continue
        cur[1].append(t.split()[0])
    blocks.append(cur)
    tot = collections.Counter()
    for name, ins in blocks:
        tot.update(ins)
        if len(ins) >= minsize:
            c = collections.Counter(ins)
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            print(f"  {name:12s} {len(ins):5d} instr, {valu:5d} VALU:", ", ".join(f"{k} {v}" for k, v in c.most_common(10)))
    print("  whole kernel:", ", ".join(f"{k} {v}" for k, v in tot.most_common(16)))
