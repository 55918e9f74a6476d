#!/bin/bash
# camera rays of BVH scenes: one tree walk per 64-path tile (k_trace_primary) against per-lane traversal (k_trace<true>): tools/r03_packet_ab.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for sc in "tests/scenes/testring.xml 1024 256" "tests/scenes/bunny.xml 1024 64"; do
  for m in 0 1 0 1; do
    echo -n "PBRT_WF_PACKET=$m  "; PBRT_WF_PACKET=$m timeout -k 10 120 python tools/run_scene.py $sc 3 2>&1 | tail -1 | cut -c1-100
  done
done
