#!/bin/bash
# Diagnostic A/B builds of libpbrt_hip.so (never shipped): each variant is timed with tools/quick_bench.py.
# usage (on the GPU box): tools/ablate.sh "<name>:<extra hipcc flags>" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT/physics-based-ray-tracing_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $flags -o /tmp/libpbrt_$name.so pbrt_api.hip 2>/dev/null || { echo "build failed: $name"; continue; }
  echo "== $name ($flags)"
  (cd $ROOT && PBRT_HIP_LIB=/tmp/libpbrt_$name.so REPS=${REPS:-3} timeout -k 5 120 python tools/quick_bench.py 2>&1 | grep -E "golden|cbox|testring|us_")
done
