#!/bin/bash
# usage: scratch/ptest.sh "<pytest -k expr>" lib1.so lib2.so ... : fused parity tests against each library build
k="$1"; shift
for l in "$@"; do
  echo "== $l"
  HOTK_LIB=$GRAFT_REPO_ROOT/$l timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 --timeout-method thread -k "$k" 2>&1 | grep -E "passed|failed|AssertionError" | head -8
done
exit 0
