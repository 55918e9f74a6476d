#!/bin/bash
# start/end of every dispatch of the last pass of a render, per queue: tools/r03_timeline.sh scene.xml res spp  (do two streams overlap?)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/timeline; mkdir -p $OUT
SCENE=$ROOT/$1; RES=$2; SPP=$3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/tools/run_scene.py $SCENE $RES $SPP 2 > $OUT/run.log 2>&1 || { tail $OUT/run.log; exit 1; }
grep "Msamples" $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "raw", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
# the last pass: from the last first-bounce k_trace on
first = max(i for i, r in enumerate(rows) if "k_trace<true" in r["Kernel_Name"] or "k_trace_primary" in r["Kernel_Name"])
while first > 0 and ("k_trace" in rows[first-1]["Kernel_Name"]) and int(rows[first]["Start_Timestamp"]) - int(rows[first-1]["Start_Timestamp"]) < 30e6: first -= 1
rows = rows[first:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"q{r.get('Queue_Id','?'):>3s} {r['Kernel_Name'][:34]:34s} grid {int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):6d}  {s/1e6:8.3f} -> {e/1e6:8.3f} ms  ({(e-s)/1e6:7.3f})")
print("span ms", (max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e6)
PY
