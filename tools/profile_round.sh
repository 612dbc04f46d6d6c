#!/bin/bash
# One mode's full profile of the round, on the GPU box:  tools/profile_round.sh <tag> <mode> [bench args...]
#   -> gpurun_out/prof_<tag>/{trace,pmc_*}, plus the summaries that get copied into profiles/:
#      gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_summary.txt, pmc_busy.json, pmc_traffic.json, bench_line.json}
set -u
TAG=$1
MODE=${2:-infer}
shift; shift || true
bash tools/profile_bench.sh "$TAG" "$MODE" "$@"
OUT=gpurun_out/prof_$TAG
COMMIT=$(cat .commit_id 2>/dev/null || echo unknown)
CMD=$(cat $OUT/command.txt)
# the source files the profiled library was built from: bench.py quotes a figure of these files only while they match
DIG=$(python3 -m ammcnet_aaai2021_amd.build --digests)
python3 tools/pmc_summary.py --json $OUT/pmc_busy.json --workload "$CMD" --commit "$COMMIT" --digests "$DIG" \
  $(find $OUT/pmc_sq $OUT/pmc_lds -name "*counter_collection.csv") > $OUT/pmc_summary.txt
PREC=s16
case " $* " in *"--precision fp32"*) PREC=fp32;; esac
python3 tools/pmc_traffic.py $OUT $PREC --mode $MODE --commit "$COMMIT" --digests "$DIG" > $OUT/pmc_traffic.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep -h '"metric"' $OUT/trace.log > $OUT/bench_line.json
tail -25 $OUT/pmc_summary.txt
