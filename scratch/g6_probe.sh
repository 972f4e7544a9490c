#!/bin/bash
# nn16 / two-phase NN kernels under the timing probes
cd $GRAFT_REPO_ROOT
for pipe in 0 1; do
for l in libhotk libhotk_noprod libhotk_noload libhotk_nopl; do
  echo "== HOTK_G6_PIPE=$pipe $l"
  HOTK_G6_PIPE=$pipe HOTK_LIB=$GRAFT_REPO_ROOT/hot_amd/$l.so timeout -k 10 100 python scratch/bench_gemm6_nn.py 2>&1 | grep "512,512\|256,320" || exit 1
done
done
