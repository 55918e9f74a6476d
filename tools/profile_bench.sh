#!/bin/bash
# rocprofv3 passes for the headline bench (run on the GPU box through gpurun).
#   1. --kernel-trace --stats : per-kernel durations
#   2. --pmc FETCH_SIZE       : HBM read bytes   (gfx950: x2 for wide streaming reads, MI355X_MICROARCH.md "HBM")
#   3. --pmc WRITE_SIZE       : HBM write bytes
# Summaries land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py condenses them for profiles/.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
python3 $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
