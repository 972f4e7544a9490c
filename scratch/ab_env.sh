#!/bin/bash
# usage: scratch/ab_env.sh VAR v1 v2 ... : bench.py (10 steps) once per value of an environment switch, same box
var=$1; shift
for v in "$@"; do
  env_line="$var=$v"
  out=$(export $env_line; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || exit 1
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['kernel_census_ms_per_step']; print('$env_line', round(d['ms_per_step'],2), {k: round(v,2) for k,v in c.items() if k in ('hotk_linear_bwd_fused','hotk_thin_gemm_bias','hotk_seg_sum_rows')})"
done
