#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the kernels' own access patterns, two rocprofv3 --pmc passes per probe (the two counters
# do not fit one pass):
#   tools/state_copy_probe.hip     k_bounce's tiled 4-B-per-lane state rows, and the 16-B-per-lane streaming copy as the control
#   tools/shade_pattern_probe.hip  k_trace / k_shade: float4 planes, 16-B records gathered by sorted slot lists and by a permutation
# -> gpurun_out/pmc_calibrate/calibration.json (copy it to profiles/rNN_pmc_calibration.json; tools/update_profiles.py cites it)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_calibrate; rm -rf $OUT; mkdir -p $OUT
for p in state_copy_probe shade_pattern_probe; do
  hipcc --offload-arch=gfx950 -O3 -o $OUT/$p $ROOT/tools/$p.hip || exit 1
done
cd /tmp && export TMPDIR=/tmp
for p in state_copy_probe shade_pattern_probe; do
  $OUT/$p > $OUT/$p.plain.log 2>&1 || { cat $OUT/$p.plain.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$p.fetch -- $OUT/$p > $OUT/$p.fetch.log 2>&1 || { tail $OUT/$p.fetch.log; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$p.write -- $OUT/$p > $OUT/$p.write.log 2>&1 || { tail $OUT/$p.write.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
def counters(probe, sub, ctr):
    """[(kernel, value in KiB)] in dispatch order"""
    f = max(glob.glob(os.path.join(out, f"{probe}.{sub}", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])) for r in rows]
res = {}
# ---- probe 1: known bytes each way, per kernel
plain = open(os.path.join(out, "state_copy_probe.plain.log")).read().splitlines()
known = int([l for l in plain if l.startswith("KNOWN_BYTES")][0].split()[1])
sec = {"known_bytes_each_way": known, "plain_run": [l.strip() for l in plain if "GB/s" in l]}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    per = {}
    for k, v in counters("state_copy_probe", sub, ctr):
        per.setdefault(k, []).append(v)
    for k, v in per.items():
        if not k.startswith("k_"): continue
        kib = sum(v) / len(v)
        sec.setdefault(k, {})[ctr + "_kib_per_dispatch"] = kib
        sec[k][ctr + "_factor_known_over_counter"] = known / (kib * 1024.0)
res["state_rows_and_wide_copy"] = sec
# ---- probe 2: patterns in dispatch order, three repetitions each
plain = [l.split() for l in open(os.path.join(out, "shade_pattern_probe.plain.log")) if l.startswith("PATTERN")]
pats = [dict(name=p[1], useful_read=int(p[3]), useful_write=int(p[5]), best_ms=float(p[7]), useful_GBps=float(p[9])) for p in plain]
for sub, ctr, key in (("fetch", "FETCH_SIZE", "useful_read"), ("write", "WRITE_SIZE", "useful_write")):
    vals = [v for k, v in counters("shade_pattern_probe", sub, ctr) if k.startswith("k_")]
    assert len(vals) == 3 * len(pats), (len(vals), len(pats))
    for i, p in enumerate(pats):
        kib = sum(vals[3 * i:3 * i + 3]) / 3.0
        p[ctr + "_kib_per_dispatch"] = kib
        p[ctr + "_factor_useful_over_counter"] = p[key] / (kib * 1024.0)
        n_rec = (p["useful_read"] // 20) if "gather" in p["name"] else 0
        if n_rec and ctr == "FETCH_SIZE":
            idx_kib = n_rec * 4 / 2 / 1024.0   # the index list streams (4 B per lane): counted at one half like any coalesced read
            p["counter_bytes_per_gathered_record"] = (kib - idx_kib) * 1024.0 / n_rec
res["stream_kernel_patterns"] = pats
json.dump(res, open(os.path.join(out, "calibration.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
