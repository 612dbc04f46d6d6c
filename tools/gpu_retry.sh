#!/bin/bash
# tools/gpu.sh, retried while the pod's GPU slots are busy (gpurun exit 3 / "status=transient": nothing charged)
#   tools/gpu_retry.sh <timeout-seconds> '<command>' [max tries]
cd "$(dirname "$0")/.."
n=${3:-8}
for i in $(seq 1 $n); do
  out=$(tools/gpu.sh "$1" "$2" 2>&1)
  echo "$out"
  echo "$out" | grep -q "status=transient" || exit 0
  echo "[gpu_retry] try $i: slots busy, waiting 60 s"
  sleep 60
done
exit 3
