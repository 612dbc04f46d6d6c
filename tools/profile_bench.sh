#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + separate PMC passes of ONE bench command.
# Usage: tools/profile_bench.sh <tag> [mode] [extra bench args...]   -> gpurun_out/prof_<tag>/...
#   mode = infer (default) | train | stress.  The program goes directly after `--` (python3 bench.py ...): no shell /
#   env hop under rocprofv3; --pmc passes are their own runs, never combined with tracing.
set -u
TAG=${1:-r02}
MODE=${2:-infer}
shift; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# per-kernel statistics of the training step: one HIP stream (kernels of the two network streams running concurrently
# stretch each other's durations; exported here, in the shell, not as an `env` hop under rocprofv3)
case "$MODE" in train|train_gan) export AMMC_TWO_STREAMS=0;; infer) export AMMC_EVAL_LANES=0;; esac
ARGS="--mode $MODE --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-grad-check $*"
echo "python3 bench.py $ARGS" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o bench -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o bench -- python3 bench.py $ARGS > $OUT/pmc_lds.log 2>&1
find $OUT -name "*.csv" | head -20
grep -h '"metric"' $OUT/trace.log | cut -c1-300
