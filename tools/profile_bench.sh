#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + separate PMC passes of the same bench command.
# Usage: tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o bench -- $CMD > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -20
grep -h '"metric"' $OUT/trace.log | cut -c1-200
