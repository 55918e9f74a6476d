#!/bin/bash
# per-launch trace + SQ counters of a scene: tools/r03_trace.sh tag scene res spp
TAG=$1; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/r03_$TAG; mkdir -p $OUT; cd $ROOT
bash tools/trace_scene.sh $2 $3 $4 2>&1 | tee $OUT/trace.txt | tail -40
bash tools/pmc_scene.sh ${TAG}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA" $2 $3 $4 2>&1 | tee $OUT/pmc_a.txt | grep "k_trace\|k_shade\|k_bounce"
bash tools/pmc_scene.sh ${TAG}_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVES" $2 $3 $4 2>&1 | tee $OUT/pmc_b.txt | grep "k_trace\|k_shade\|k_bounce"
