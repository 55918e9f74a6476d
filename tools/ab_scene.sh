#!/bin/bash
# A/B of prebuilt libraries on one scene: tools/ab_scene.sh "scene.xml res spp" name=lib.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
ARGS=$1; shift
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    echo "== $name (round $round)"
    PBRT_HIP_LIB=$ROOT/$lib timeout -k 5 120 python tools/run_scene.py $ARGS 3 2>&1 | grep Msamples
  done
done
