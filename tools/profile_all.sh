#!/bin/bash
# the rocprofv3 passes of tools/profile_bench.sh over every bench config (run on the GPU box):  tools/profile_all.sh <round tag, e.g. r04>
# then, in the build container:  for c in cbox us_sphere_box testring us_testring cbox4k; do python tools/update_profiles.py <tag>_$c $c; done
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; R=${1:-r04}
for c in cbox us_sphere_box testring us_testring cbox4k; do
  echo "=== $c"; bash $ROOT/tools/profile_bench.sh ${R}_$c --config $c > $ROOT/gpurun_out/prof_${R}_$c.log 2>&1 || { tail -20 $ROOT/gpurun_out/prof_${R}_$c.log; exit 1; }
  tail -4 $ROOT/gpurun_out/prof_${R}_$c.log
done
