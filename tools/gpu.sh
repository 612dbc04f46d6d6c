#!/bin/bash
# local wrapper: stamp the snapshot with the commit it was taken at, then hand the command to gpurun
#   tools/gpu.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.."
id=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!PROGRESS.jsonl' || id="$id+dirty"
echo "$id" > .commit_id
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
