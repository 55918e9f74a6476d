#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel-trace stats + PMC passes) into a small text/JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    n = name.split("(")[0]
    for k in ("k_bounce", "k_trace", "k_shade", "k_us_bounce", "k_us_shade", "k_us_first", "k_film_accum", "k_film_resolve", "k_scale"):
        if k in n:
            return n[n.index(k):][:60]
    return n[:60]


res = {}
st = find("trace", "*kernel_stats.csv")
if st:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    rows = list(csv.DictReader(open(st)))
    for r in rows[:12]:
        print(f"{short(r['Name']):50s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>14s} avg_ns={float(r['AverageNs']):>12.0f} pct={r['Percentage']}")
    res["kernel_stats"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]), "total_ns": int(r["TotalDurationNs"]),
                            "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])} for r in rows[:12]]
kt = find("trace", "*kernel_trace.csv")
if kt:
    agg = defaultdict(lambda: [0, 0.0, 0, 0, 0])
    for r in csv.DictReader(open(kt)):
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[2] = int(r.get("VGPR_Count", 0) or 0)
        a[3] = int(r.get("SGPR_Count", 0) or 0)
        a[4] = int(r.get("LDS_Block_Size", 0) or 0)
    print("== per-kernel (kernel_trace.csv) ==")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{k:50s} n={a[0]:6d} avg_us={a[1] / a[0] / 1e3:10.2f} vgpr={a[2]} sgpr={a[3]} lds={a[4]}")
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(tag, "*counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    print(f"== {ctr} (sum over dispatches; unit of the counter: KiB) ==")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"{k:50s} n={a[0]:6d} total={a[1]:.4g} per_dispatch={a[1] / a[0]:.6g}")
    res[ctr] = {k: {"dispatches": a[0], "total_kib": a[1], "per_dispatch_kib": a[1] / a[0]} for k, a in agg.items()}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
