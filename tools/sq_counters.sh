#!/bin/bash
# SQ counters of a scene render, two rocprofv3 --pmc passes (issue / wait counters, then LDS / lane counters), per kernel:
#   tools/sq_counters.sh <tag> scene.xml res spp   -> gpurun_out/<tag>_sq_counters.txt (+ the derived ratios of the k_trace kernels)
set -o pipefail
TAG=$1; SCENE=$2; RES=$3; SPP=$4
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${TAG}_sq_counters.txt
cd $ROOT
bash tools/pmc_scene.sh ${TAG}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA" $SCENE $RES $SPP > $OUT 2>&1 || { cat $OUT; exit 1; }
bash tools/pmc_scene.sh ${TAG}_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVES" $SCENE $RES $SPP >> $OUT 2>&1 || { cat $OUT; exit 1; }
python3 - $OUT <<'PY' | tee -a $OUT
import re, sys, collections
agg = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    if " | " not in l: continue
    k, rest = l.split(" | ", 1)
    for m in re.finditer(r"(\w+)=([0-9.e+]+)\(n=(\d+)\)", rest): agg[k.strip()][m.group(1)] = float(m.group(2))
print("--- derived")
for k, d in agg.items():
    if not all(c in d for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU")): continue
    lds = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"] if d.get("SQ_LDS_IDX_ACTIVE") else 0.0
    print(f"{k:36s} valu_busy {d['SQ_INSTS_VALU'] / (16 * d['SQ_BUSY_CYCLES']):.3f}  lanes {d['SQ_THREAD_CYCLES_VALU'] / (64 * d['SQ_ACTIVE_INST_VALU']):.3f}  "
          f"sca/valu {d['SQ_ACTIVE_INST_SCA'] / d['SQ_ACTIVE_INST_VALU']:.3f}  lds_conflict {lds:.3f}  wait_any {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.3f}")
PY
