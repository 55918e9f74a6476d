#!/bin/bash
# one rocprofv3 --pmc pass over an arbitrary python script: tools/pmc_cmd.sh "<CTRS>" script.py [args]; prints per-kernel sums
set -o pipefail
CTRS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_cmd; rm -rf $OUT; mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/raw -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
f = max(glob.glob(os.path.join(sys.argv[1], "raw", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "k_" in k: print(k + " | " + " ".join(f"{c}={v:.4g}(n={n[(k,c)]})" for c, v in sorted(d.items())))
PY
