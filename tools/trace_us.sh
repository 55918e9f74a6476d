#!/bin/bash
# per-dispatch kernel durations of one ultrasound acquisition (us_sphere_box, 5 x 64 x PPR paths)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/trace_us; mkdir -p $OUT
cat > /tmp/us_run.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import pbrt_amd as mi
us = mi.load_file(os.environ.get("US_SCENE", "$ROOT/tests/scenes/us_sphere_box.xml"))
ui = us.integrator()
for i in range(2):
    ui._acquire(us, ui.quirks, paths_per_ray=int(os.environ.get("PPR", "65536")))
st = mi.default_context().stats()
print("kernel_ms", st["kernel_ms"], "live", st["live"][:12], "segments", st["segments"], "samples", st["samples"])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 /tmp/us_run.py > $OUT/run.log 2>&1 || { tail $OUT/run.log; exit 1; }
tail -1 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "raw", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_us" in r["Kernel_Name"] or "k_trace" in r["Kernel_Name"] or "k_scale" in r["Kernel_Name"] or "k_reduce" in r["Kernel_Name"]]
rows = rows[len(rows)//2:]
t0 = int(rows[0]["Start_Timestamp"]); prev = None
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{i:3d} {r['Kernel_Name'][:34]:34s} start {(s-t0)/1e3:8.1f} dur {(e-s)/1e3:8.1f} gap {((s-prev)/1e3 if prev else 0):6.1f}")
    prev = e
print("span us", (int(rows[-1]["End_Timestamp"]) - t0) / 1e3)
PY
