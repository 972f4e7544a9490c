#!/bin/bash
# fused parity tests, then (only if green) per-kernel times of the core micro-benchmark
k="$1"; n=${2:-10e6}
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "$k" > gpurun_out/t1.log 2>&1
rc=$?
tail -${3:-6} gpurun_out/t1.log
[ $rc -eq 0 ] || exit $rc
scratch/kfused.sh kf $n
