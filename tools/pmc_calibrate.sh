#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on k_bounce's own state access pattern (tools/state_copy_probe.hip): two rocprofv3
# --pmc passes (the two counters do not fit one pass), result in gpurun_out/pmc_calibrate/calibration.json
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_calibrate; rm -rf $OUT; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o $OUT/state_copy_probe $ROOT/tools/state_copy_probe.hip || exit 1
cd /tmp && export TMPDIR=/tmp
$OUT/state_copy_probe > $OUT/plain.log 2>&1 || { cat $OUT/plain.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $OUT/state_copy_probe > $OUT/fetch.log 2>&1 || { tail $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $OUT/state_copy_probe > $OUT/write.log 2>&1 || { tail $OUT/write.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
known = int([l for l in open(os.path.join(out, "plain.log")) if l.startswith("KNOWN_BYTES")][0].split()[1])
res = {"known_bytes_each_way": known, "plain_run": [l.strip() for l in open(os.path.join(out, "plain.log")) if "GB/s" in l]}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = max(glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        kib = sum(v) / len(v)
        res.setdefault(k, {})[ctr + "_kib_per_dispatch"] = kib
        res[k][ctr + "_factor_known_over_counter"] = known / (kib * 1024.0)
json.dump(res, open(os.path.join(out, "calibration.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
