#!/bin/bash
# after `tools/final_round.sh <tag>` came back: copy the judged summaries from gpurun_out/prof_<tag>_*/ into profiles/
#   tools/install_profiles.sh <tag>
cd "$(dirname "$0")/.."
TAG=$1
for m in infer stress train gan fp32; do
  [ -d gpurun_out/prof_${TAG}_$m ] || continue
  for f in kernel_stats.csv pmc_summary.txt pmc_busy.json pmc_traffic.json bench_line.json; do
    cp gpurun_out/prof_${TAG}_$m/$f profiles/${TAG}_${m}_$f
  done
done
[ -s gpurun_out/${TAG}_final_bench_line.json ] && cp gpurun_out/${TAG}_final_bench_line.json profiles/${TAG}_final_bench_line.json
ls profiles | grep -c "^${TAG}_"
