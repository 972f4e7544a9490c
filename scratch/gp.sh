#!/bin/bash
# retry wrapper: scratch/gp.sh <timeout> <command...>
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  out=$(/usr/local/graft/bin/gpurun --timeout $t -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 40; else echo "$out" | tail -${TAILN:-25}; exit 0; fi
done
echo "no box after retries"
