#!/bin/bash
# CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (GPU ASan is not available on this pool):
# builds oracle/_build/liboracle_asan.so and runs the oracle-only test files against it.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT && make -C oracle asan || exit 1
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
PBRT_ORACLE_LIB=$ROOT/oracle/_build/liboracle_asan.so \
python -m pytest tests/test_oracle_transport.py tests/test_oracle_cone.py tests/test_known_answers.py tests/test_parallel_gloo.py tests/test_pinned_transcription.py -x -q -m "not gpu"
