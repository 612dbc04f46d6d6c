#!/bin/bash
# kernel trace (+stats) of an arbitrary python command on the GPU box:  tools/profile_cmd.sh <tag> <script> [args...]
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 "$@" > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log | cut -c1-300
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-100s calls %5s total %9.3f ms avg %9.1f us %6s%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                  float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
