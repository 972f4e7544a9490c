#!/bin/bash
# usage: scratch/gtest.sh "<pytest -k expr>" [entries]: fused parity tests, then (only if green) the core micro-benchmark
k="$1"; n=${2:-10e6}
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "$k" > gpurun_out/t1.log 2>&1
rc=$?
tail -25 gpurun_out/t1.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python scratch/bench_fused.py $n 3 || exit 1
HOTK_CORE=valu timeout -k 10 200 python scratch/bench_fused.py $n 3
