#!/bin/bash
# per-dispatch kernel durations of one bench step, grouped by position in the pass (= depth)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/trace_depth; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/run.log 2>&1 || { tail $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "raw", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_bounce" in r["Kernel_Name"] or "k_film" in r["Kernel_Name"]]
rows = rows[len(rows)//2:]   # second (timed) step
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
acc = {}
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = "film_accum" if "film_accum" in r["Kernel_Name"] else ("resolve" if "resolve" in r["Kernel_Name"] else ("b_first" if "<true" in r["Kernel_Name"] else "b"))
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    if i < 8: print(f"{i:3d} {name:10s} start {(s - t0) / 1e3:9.1f} us dur {(e - s) / 1e3:8.1f} us gap {gap:6.1f}")
    pos = i % 7
    a = acc.setdefault(pos, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += gap
for pos, a in sorted(acc.items()):
    print(f"pos {pos}: n={a[0]} avg dur {a[1] / a[0]:8.1f} us avg gap {a[2] / a[0]:6.1f} us")
print("total span us", (int(rows[-1]["End_Timestamp"]) - t0) / 1e3)
PY
