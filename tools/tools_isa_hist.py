#!/usr/bin/env python3
"""Instruction histogram per kernel of a device-only assembly listing (hipcc --offload-device-only -S).
usage: tools_isa_hist.py file.s [kernel-name-substring]"""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
hist = collections.defaultdict(collections.Counter)
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        continue
    if cur is None:
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
        continue
    m = re.match(r"^\t([a-z_0-9]+)\s", line)
    if m and not m.group(1).startswith("."):
        hist[cur][m.group(1)] += 1
for k, h in hist.items():
    if want not in k:
        continue
    tot = sum(h.values())
    valu = sum(v for i, v in h.items() if i.startswith("v_"))
    print(f"== {k[:110]}\n   total {tot}  valu {valu}  salu {sum(v for i, v in h.items() if i.startswith('s_'))}  ds {sum(v for i, v in h.items() if i.startswith('ds_'))}  "
          f"vmem {sum(v for i, v in h.items() if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))}")
    print("   " + ", ".join(f"{i} {v}" for i, v in h.most_common(40)))
