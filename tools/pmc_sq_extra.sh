#!/bin/bash
# extra SQ counter passes of the bench (LDS / VALU / VMEM issue activity):  tools/pmc_sq_extra.sh <tag>
set -u
TAG=${1:-x}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-grad-check"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_lds -o bench -- $CMD > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc_valu -o bench -- $CMD > $OUT/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_any -o bench -- $CMD > $OUT/pmc_any.log 2>&1
python3 tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv") | grep -E "${2:-conv_tap_s16_kernel<4, 1, 2, 4}"
