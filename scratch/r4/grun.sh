#!/bin/bash
# grun.sh TIMEOUT 'command' : gpurun with retries while no GPU slot is free (exit code 3 = nothing charged)
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/grun_last.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then cat /tmp/grun_last.log; exit $rc; fi
  sleep 45
done
cat /tmp/grun_last.log; exit 3
