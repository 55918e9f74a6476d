#!/bin/bash
# usage: tools/pmc.sh <tag> "<CTR1 CTR2 ...>" [bench args]   -- one rocprofv3 --pmc pass over bench.py
set -o pipefail
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "raw", "**", "*counter_collection.csv"), recursive=True)[0]
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
with open(os.path.join(out, "summary.txt"), "w") as o:
    for k, d in agg.items():
        if "k_" not in k: continue
        line = k + " | " + " ".join(f"{c}={v:.4g}(n={n[(k,c)]})" for c, v in sorted(d.items()))
        print(line); o.write(line + "\n")
PY
