#!/bin/bash
# GPU tests + the ring scene timing (tools/gpu_check.sh [tag])
set -o pipefail
TAG=${1:-chk}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/chk_$TAG; mkdir -p $OUT
cd $ROOT
# the first BVH radiance test alone, with a short leash (a hung kernel must not sit there for minutes)
timeout -k 5 90 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "ring_meshes_small" 2>&1 | tee $OUT/pytest_first.log | tail -25; rc=${PIPESTATUS[0]}; echo "first rc $rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest.log | tail -15; rc=${PIPESTATUS[0]}; echo "pytest rc $rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python tools/run_scene.py tests/scenes/testring.xml 1024 64 3 2>&1 | tail -2
timeout -k 10 120 python tools/run_scene.py tests/scenes/simple.xml 512 64 3 2>&1 | tail -1
