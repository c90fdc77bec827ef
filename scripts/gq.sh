#!/bin/bash
# gpurun with re-queueing when no GPU slot is free (exit code 3: nothing ran, nothing charged).
# usage: scripts/gq.sh <timeout_s> '<command>'   - any other exit code is final.
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
