#!/bin/bash
# End-of-round validation on the GPU box:  tools/final_round.sh <tag>   (e.g. r05d)
#   1. the whole GPU test suite, 2. the profile set of every bench mode at HEAD (tools/profile_round.sh), installed into
#   profiles/ of the box's copy so that 3. the full `python bench.py` line quotes them (bench.py: profile_is_current).
# Everything lands under gpurun_out/ (final_<tag>_*.log, prof_<tag>_*/, <tag>_final_bench_line.json).
set -u
TAG=$1
python -m pytest tests -q -m gpu -x > gpurun_out/final_${TAG}_tests.log 2>&1
tail -3 gpurun_out/final_${TAG}_tests.log
for mode in infer stress train train_gan; do
  short=$mode; [ $mode = train_gan ] && short=gan
  bash tools/profile_round.sh ${TAG}_$short $mode > gpurun_out/final_${TAG}_prof_$short.log 2>&1
  for f in kernel_stats.csv pmc_summary.txt pmc_busy.json pmc_traffic.json bench_line.json; do
    cp gpurun_out/prof_${TAG}_$short/$f profiles/${TAG}_${short}_$f
  done
  # the raw counter CSVs stay on the box (only the summaries are judged): keep the merge small
  rm -rf gpurun_out/prof_${TAG}_$short/pmc_sq gpurun_out/prof_${TAG}_$short/pmc_lds gpurun_out/prof_${TAG}_$short/pmc_fetch gpurun_out/prof_${TAG}_$short/pmc_write gpurun_out/prof_${TAG}_$short/trace
done
# the exact-fp32 kernels (the S16 guard's fallback; bench.py's `fp32_exact` leg quotes rNN_fp32_pmc_traffic.json)
bash tools/profile_round.sh ${TAG}_fp32 infer --precision fp32 > gpurun_out/final_${TAG}_prof_fp32.log 2>&1
for f in kernel_stats.csv pmc_summary.txt pmc_busy.json pmc_traffic.json bench_line.json; do
  cp gpurun_out/prof_${TAG}_fp32/$f profiles/${TAG}_fp32_$f
done
rm -rf gpurun_out/prof_${TAG}_fp32/pmc_sq gpurun_out/prof_${TAG}_fp32/pmc_lds gpurun_out/prof_${TAG}_fp32/pmc_fetch gpurun_out/prof_${TAG}_fp32/pmc_write gpurun_out/prof_${TAG}_fp32/trace
# (the box's profiles/ does not travel back: tools/install_profiles.sh <tag> copies the summaries from gpurun_out/ at home)
python bench.py > gpurun_out/${TAG}_final_bench_line.json 2> gpurun_out/final_${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_final_bench_line.json
