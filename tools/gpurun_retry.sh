#!/bin/bash
# usage: gpurun_retry.sh <timeout> '<command>'  -- retries ONLY when no box is free (exit 3)
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[retry] no box free (try $i), sleeping 150s"
  sleep 150
done
exit 3
