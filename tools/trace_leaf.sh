#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/leaf -- python3 $ROOT/tools/leaf_rays.py > $ROOT/gpurun_out/leaf.log 2>&1
grep prims $ROOT/gpurun_out/leaf.log
python3 - "$ROOT" <<'PY'
import csv, glob, sys, os
f = max(glob.glob(sys.argv[1] + "/gpurun_out/leaf/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if "k_ray" in r["Kernel_Name"]: print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
