#!/usr/bin/env python3
"""Per-launch durations of the cbox bench grouped by their position in a pass (= depth), from the kernel trace that
tools/profile_bench.sh leaves in gpurun_out/prof_<tag>/trace.  usage: tools/per_depth.py gpurun_out/prof_r02_cbox > profiles/r02_cbox_per_depth.txt"""
import csv, glob, json, os, sys
out = sys.argv[1]
f = max(glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_bounce" in r["Kernel_Name"] or "k_film_accum" in r["Kernel_Name"]]
passes, cur = [], []
for r in rows:
    if "k_bounce<true" in r["Kernel_Name"] and cur:
        passes.append(cur); cur = []
    cur.append(r)
passes.append(cur)
lens = [len(p) for p in passes]
n = max(set(lens), key=lens.count)   # the steady state (the set-up render of bench.py learns the launch plan)
passes = [p for p in passes if len(p) == n]
live = None
bj = os.path.join(out, "bench_under_rocprof.json")
print(f"cbox 512^2 x 256 spp, default launch plan; rocprofv3 --kernel-trace of tools/profile_bench.sh ({len(passes)} passes traced)")
print("position in the pass | kernel | mean us | min..max us")
tot = 0.0
for k in range(n):
    d = [(int(p[k]["End_Timestamp"]) - int(p[k]["Start_Timestamp"])) / 1e3 for p in passes]
    name = passes[0][k]["Kernel_Name"].split("(")[0].replace("void ", "")
    print(f"{k} | {name:28s} | {sum(d) / len(d):8.1f} | {min(d):8.1f}..{max(d):8.1f}")
    tot += sum(d) / len(d)
print(f"sum per pass {tot:.1f} us")
