#!/usr/bin/env python3
"""Instruction histogram of the k-loop (the span between a kernel's first and last MFMA) from the
-save-temps assembly (`make -C transcoder-sae-circuits_amd/csrc asm`).  usage: loop_hist.py <substring of kernel name>"""
import os
import sys
from collections import Counter

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "transcoder-sae-circuits_amd", "csrc", "build",
                    "asm", "encode_f32-hip-amdgcn-amd-amdhsa-gfx950.s")
s = open(path).read()
key = sys.argv[1]
i = s.index(key)
i = s.index(":", i)
j = s.index("s_endpgm", i)
body = s[i:j].split("\n")
idx = [k for k, l in enumerate(body) if "v_mfma" in l]
c = Counter()
for l in body[idx[0]:idx[-1] + 1]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    c[l.split()[0]] += 1
print(len(idx), "MFMAs;", sum(c.values()) - len(idx), "other instructions")
for k, v in c.most_common():
    print(f"{v:5d} {k}")
