#!/bin/bash
# one launch per bounce (PBRT_USQ_NO_FUSED_BOUNCES) of ONE 16 Mi-path pass of an ultrasound acquisition, per-launch durations from a kernel trace:
#   US_SCENE_KW="primary_rays=emitter" tools/us_per_depth.sh [scene.xml]     (what does bounce d cost with and without emitter rays?)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/us_per_depth; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/us_depth_run.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import pbrt_amd as mi
kw = dict(kv.split("=") for kv in os.environ.get("US_SCENE_KW", "").split(";") if kv)
us = mi.load_file("$ROOT/" + (sys.argv[1] if len(sys.argv) > 1 else "tests/scenes/us_sphere_box.xml"), **kw)
ui = us.integrator()
q = ui.quirks | mi._capi.USQ_NO_FUSED_BOUNCES
for i in range(2):
    ui._acquire(us, q, paths_per_ray=52429)
st = mi.default_context().stats()
print("kernel_ms", st["kernel_ms"], "live", st["live"][:12], "segments", st["segments"], "samples", st["samples"])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 /tmp/us_depth_run.py "$@" > $OUT/run.log 2>&1 || { tail $OUT/run.log; exit 1; }
tail -1 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "raw", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_us" in r["Kernel_Name"]]
rows = rows[len(rows)//2:]
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{i:3d} {r['Kernel_Name'][:44]:44s} dur {(e-s)/1e3:8.1f} us")
PY
