#!/bin/bash
# usage: tools/_gpu.sh NAME TIMEOUT 'command'   -- waits for a free slot (exit 3 = nothing charged), then runs once
name=$1; to=$2; shift 2
for i in $(seq 1 40); do
  gpurun --timeout $to -- "$@" > gpurun_out/${name}_call.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> gpurun_out/${name}_call.log; echo finished >> gpurun_out/${name}_call.log; exit $rc; fi
  sleep 75
done
echo "gave up" >> gpurun_out/${name}_call.log
