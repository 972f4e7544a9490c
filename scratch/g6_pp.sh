#!/bin/bash
cd $GRAFT_REPO_ROOT
for pipe in 2 0; do
  echo "== HOTK_G6_PIPE=$pipe"
  HOTK_G6_PIPE=$pipe timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -k "gemm6_products" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
  HOTK_G6_PIPE=$pipe timeout -k 10 100 python scratch/bench_gemm6_nn.py 2>&1 | grep "nn " || exit 1
done
