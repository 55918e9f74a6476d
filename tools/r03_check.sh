#!/bin/bash
# GPU tests + the ring scene timing (tools/r03_check.sh [tag])
set -o pipefail
TAG=${1:-chk}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/r03_$TAG; mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -15 $OUT/pytest.log; echo "pytest rc $rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python tools/run_scene.py tests/scenes/testring.xml 1024 64 3 2>&1 | tail -2
timeout -k 10 120 python tools/run_scene.py tests/scenes/simple.xml 512 64 3 2>&1 | tail -1
