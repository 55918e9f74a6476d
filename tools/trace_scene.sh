#!/bin/bash
# per-dispatch kernel durations of the last render of tools/run_scene.py: tools/trace_scene.sh scene.xml res spp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/trace_scene; mkdir -p $OUT
SCENE=$ROOT/$1; RES=$2; SPP=$3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/tools/run_scene.py $SCENE $RES $SPP 2 > $OUT/run.log 2>&1 || { tail $OUT/run.log; exit 1; }
grep "Msamples" $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
f = max(glob.glob(os.path.join(sys.argv[1], "raw", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
rows = rows[len(rows)//2:]
t0 = int(rows[0]["Start_Timestamp"]); prev = None
agg = collections.OrderedDict()
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if i < 10: print(f"{i:3d} {r['Kernel_Name'][:40]:40s} dur {(e-s)/1e3:9.1f} us gap {((s-prev)/1e3 if prev else 0):6.1f}")
    prev = e
    k = r["Kernel_Name"][:40]; a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
for k, a in agg.items(): print(f"{k:40s} n={a[0]:4d} total {a[1]:10.1f} us avg {a[1]/a[0]:9.1f}")
print("span us", (int(rows[-1]["End_Timestamp"]) - t0) / 1e3)
PY
