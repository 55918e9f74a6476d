#!/bin/bash
# round-3 baseline on one box: GPU tests, the four bench configs, per-launch trace and SQ wait counters of the ring scene
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/r03_base; mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
for c in cbox testring us_sphere_box cbox4k; do
  python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err || { tail -5 $OUT/bench_$c.err; exit 1; }
  python - $OUT/bench_$c.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(d['metric'][:50], d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])
PY
done
bash tools/trace_scene.sh tests/scenes/testring.xml 1024 64 > $OUT/trace_ring.txt 2>&1; tail -25 $OUT/trace_ring.txt
bash tools/pmc_scene.sh r03_ring_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" tests/scenes/testring.xml 1024 64 > $OUT/pmc_ring_a.txt 2>&1; cat $OUT/pmc_ring_a.txt
bash tools/pmc_scene.sh r03_ring_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA" tests/scenes/testring.xml 1024 64 > $OUT/pmc_ring_b.txt 2>&1; cat $OUT/pmc_ring_b.txt
