#!/bin/bash
# Cornell box (BASELINE config 2) with and without the two-tiles-per-wave kernel (k_chain_pair, diagnostic build): tools/r03_pair_ab.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
export PBRT_HIP_LIB=$ROOT/physics-based-ray-tracing_amd/csrc/libpbrt_hip_diag.so
for m in 0 4 3 5 0 4; do
  echo -n "PBRT_PAIR_MERGE=$m  "; PBRT_PAIR_MERGE=$m timeout -k 10 120 python tools/run_scene.py tests/scenes/cbox.xml 512 256 4 2>&1 | tail -1
done
