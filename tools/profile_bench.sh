#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).   usage: tools/profile_bench.sh <tag> [bench args]
#   1. --kernel-trace --stats : per-kernel durations
#   2. --pmc FETCH_SIZE       : HBM read bytes   (gfx950: x2 for wide streaming reads, MI355X_MICROARCH.md "HBM";
#                               the factor on k_bounce's own 4-B-per-lane row pattern: tools/pmc_calibrate.sh)
#   3. --pmc WRITE_SIZE       : HBM write bytes  (separate pass: the two counters do not fit one)
# Summaries land in gpurun_out/prof_<tag>/ ; tools/update_profiles.py <tag> <config> copies them into profiles/.
set -o pipefail
TAG=${1:-r04}; shift
# one process per profiler: bench.py --gpus N > 1 would start its ranks as children of a process whose GPU the profiler's preloaded
# library has already initialised (and the children would inherit the preload).  Profile multi-rank runs rank by rank instead.
for a in "$@"; do case "$prev$a" in --gpus[2-9]*|--gpus=[2-9]*|--gpus1[0-9]*) echo "profile_bench.sh: --gpus > 1 is refused (see the comment)"; exit 2;; esac; prev=$a; done
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also "$@" > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-also "$@" > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
# 4. --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES : VALU-issue utilisation of the dominant kernel (a wave64 VALU instruction occupies its SIMD-32 for
#    2 cycles; SQ_BUSY_CYCLES is summed over the 32 shader engines, 1024 SIMDs): busy = SQ_INSTS_VALU / (16 * SQ_BUSY_CYCLES); and the
#    share of the lanes that are active per VALU instruction: lane_active = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also "$@" > $OUT/pmc_sq.log 2>&1 || { tail -20 $OUT/pmc_sq.log; exit 1; }
python3 $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
