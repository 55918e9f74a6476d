#!/bin/bash
# A/B timing of prebuilt libraries in one GPU session: tools/ab_libs.sh name=path.so ...  (interleaved, 2 rounds)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    echo "== $name (round $round)"
    PBRT_HIP_LIB=$ROOT/$lib REPS=${REPS:-5} timeout -k 5 120 python tools/quick_bench.py 2>&1 | grep -E "golden|cbox|testring|us_"
  done
done
