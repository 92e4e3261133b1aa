#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc SQ passes of tools_sq.sh: per kernel (template arguments shortened) the mean of every
counter per launch, over the launches of the instance's steady state.  SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_*
count quad-cycles (4 shader clocks), summed over the waves of a launch; SQ_BUSY_CYCLES is per shader engine."""
import collections
import csv
import glob
import os
import re
import sys

root, tag = sys.argv[1], sys.argv[2]
args = " ".join(sys.argv[3:])


def short(name):
    n = name.split("(")[0]
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\b(wcol|wrow|wrown|wcoln|wholerow)::", "", n)
    pol = "pdr" if "PdrP" in n else "pddr" if "PddrP" in n else "admm" if "AdmmP" in n else None
    base = n.split("<")[0]
    targs = n[len(base):]
    targs = re.sub(r"(PdrP|PddrP|AdmmP)<[^>]*>", lambda m: pol, targs)
    return (base + targs)[:100]


per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(os.path.join(root, "sq*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        per[short(r["Kernel_Name"])][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])

lines = [f"SQ counters per launch (mean over launches), bench.py {args}, rocprofv3 --pmc (two passes of 8 counters, one run each)"]
for k in sorted(per, key=lambda k: -sum(per[k].get("SQ_WAVE_CYCLES", {0: 0}).values())):
    if not k.startswith("k_"):
        continue
    c = per[k]
    n = max(len(v) for v in c.values())
    if n < 3:
        continue
    m = {name: sum(v.values()) / len(v) for name, v in c.items()}
    lines.append(f"{k}  (n={n})")
    lines.append("    " + ", ".join(f"{name}={val:.4g}" for name, val in sorted(m.items())))
    d = []
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"]:
        d.append(f"VALU instructions per wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}")
        if "SQ_INSTS_LDS" in m:
            d.append(f"LDS instructions per wave {m['SQ_INSTS_LDS'] / m['SQ_WAVES']:.0f}")
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        d.append(f"VALU-active share of wave cycles {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        d.append(f"waiting share (s_waitcnt / barrier) {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        d.append(f"issue-stall share {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        d.append(f"bank-conflict share of LDS-active cycles {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
    if d:
        lines.append("    -> " + "; ".join(d))
out = os.path.join(root, f"{tag}_sq_counters.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
