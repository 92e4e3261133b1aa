#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools_pmc.sh into <tag>_pmc.json (what bench.py reads for `roofline.traffic`) and
<tag>_kernel_stats.csv.

HBM bytes per launch of a kernel = 2 x FETCH_SIZE + WRITE_SIZE, both in bytes:
  * rocprofv3 reports FETCH_SIZE and WRITE_SIZE in KiB;
  * on gfx950 FETCH_SIZE counts a 128-byte read request as 64 bytes, so reads are doubled; WRITE_SIZE is exact for the
    16-byte-per-lane stores these kernels issue.
Of every kernel family the instances launched most often are the full (consume + produce) launches of the solver
loops (two alternating ones for Chambolle-Pock: their mean is reported); the first / last launches of a loop are other
template instances and are left out."""
import collections
import csv
import glob
import json
import os
import re
import sys

root, tag = sys.argv[1], sys.argv[2]


def family(name):
    """bench.py's name of the kernel family, or None for kernels outside the per-kernel table."""
    n = name.split("(")[0]
    pol = "pdr" if "PdrP" in n else "pddr" if "PddrP" in n else "admm" if "AdmmP" in n else None
    for base, pretty in (("k_wave_rowN", "k_wave_rowN"), ("k_wave_row", "k_wave_row"), ("k_wave_cpN", "k_wave_cpN"), ("k_wave_cp", "k_wave_cp"),
                         ("k_wave_colN", "k_wave_colN"), ("k_wave_col", "k_wave_col"), ("k_long_inv_w", "k_long_inv_w"), ("k_long_fwd_w", "k_long_fwd_w"),
                         ("k_long_pix", "k_long_pix"), ("k_row", "k_row"), ("k_col", "k_col")):
        if re.search(r"\b" + base + r"\b|" + base + r"[I<]", n):
            if base.startswith("k_wave_col"):
                m = re.search(base + r"(?:ILi|<)(\d)", n)
                return f"{pretty}<{'resolvent' if m and m.group(1) == '0' else 'multiplier'}>"
            if pol and base in ("k_wave_row", "k_wave_rowN", "k_long_pix", "k_row"):
                return f"{pretty}<{pol}>"
            return pretty
    return None


def counter_per_dispatch(sub, counter):
    """{kernel name: [value per dispatch]} (rows of one dispatch, if split, are summed)."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in per.items()}


def main():
    bench = None
    try:
        bench = json.loads([l for l in open(os.path.join(root, f"{tag}_bench.json")) if l.startswith("{")][-1])
    except Exception:
        pass
    fetch, write = counter_per_dispatch("pmc_fetch", "FETCH_SIZE"), counter_per_dispatch("pmc_write", "WRITE_SIZE")
    fams = collections.defaultdict(list)
    for name in set(fetch) | set(write):
        fam = family(name)
        if fam:
            fams[fam].append(name)
    kernels = {}
    for fam, names in fams.items():
        count = lambda n: len(fetch.get(n, [])) + len(write.get(n, []))
        full = max(names, key=count)
        # the steady-state instances: the most frequent one and any other launched at least 40 % as often (Chambolle-Pock
        # alternates two row kernels); the first / last launches of a loop fall below that
        steady = [n for n in names if count(n) >= 0.4 * count(full)]
        fv = [v for n in steady for v in fetch.get(n, [])]
        wv = [v for n in steady for v in write.get(n, [])]
        if not fv or not wv:
            continue
        f_kib, w_kib = sum(fv) / len(fv), sum(wv) / len(wv)
        kernels[fam] = {"hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0, "fetch_size_kib": f_kib, "write_size_kib": w_kib,
                        "dispatches": [len(fv), len(wv)], "instance": full.split("(")[0][:160]}
    rec = {"ddn_version": (bench or {}).get("config", {}).get("ddn_version"), "size": None, "batch": (bench or {}).get("config", {}).get("images_per_gpu"),
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of the same bench.py command ({tag}); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB",
           "kernels": kernels}
    if bench:
        m = re.match(r"(\d+)x(\d+)", bench["config"].get("workload", ""))
        if m:
            rec["size"] = [int(m.group(1)), int(m.group(2))]
        rec["bench_kernel_ms"] = {k: v.get("ms") for k, v in bench.get("roofline", {}).get("kernels", {}).items()}
        rec["algorithmic_bytes_per_launch"] = {k: v.get("algorithmic_bytes") for k, v in bench.get("roofline", {}).get("kernels", {}).items()}
    json.dump(rec, open(os.path.join(root, f"{tag}_pmc.json"), "w"), indent=1)
    # per-kernel durations of the kernel-trace run
    for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w") as o:
            w = csv.writer(o)
            w.writerow(["family", "calls", "average_us", "min_us", "max_us", "percentage", "name"])
            for r in rows:
                w.writerow([family(r["Name"]) or "", r["Calls"], f"{float(r['AverageNs']) / 1e3:.2f}", f"{float(r['MinNs']) / 1e3:.2f}", f"{float(r['MaxNs']) / 1e3:.2f}",
                            r["Percentage"], r["Name"][:200]])
    for fam, k in sorted(kernels.items()):
        alg = (rec.get("algorithmic_bytes_per_launch") or {}).get(fam)
        print(f"{fam:28s} traffic {k['hbm_bytes_per_launch'] / 1e6:9.1f} MB  algorithmic {alg / 1e6 if alg else float('nan'):9.1f} MB  n={k['dispatches']}")


if __name__ == "__main__":
    main()
