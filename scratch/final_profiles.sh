#!/bin/bash
# round-3 final measurements (each step under its own limit; stops at the first failure)
set -e
cd $GRAFT_REPO_ROOT
bash scratch/kstats.sh r3f_c1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-precondition > /dev/null
tail -1 gpurun_out/r3f_c1.log > gpurun_out/r3f_c1_profiled_line.json
timeout -k 10 400 python3 bench.py > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err
echo "bench done"
bash scratch/pmc_traffic.sh r3f_pmc 10e6 > /dev/null
echo "pmc done"
timeout -k 10 200 python3 bench.py --config c2 --molecules 128 > gpurun_out/r3f_c2_128.json 2> gpurun_out/r3f_c2_128.err
timeout -k 10 200 python3 bench.py --config c2 --molecules 4096 > gpurun_out/r3f_c2_4096.json 2> gpurun_out/r3f_c2_4096.err
echo "c2 done"
timeout -k 10 300 python3 bench.py --storage bf16 --no-cpu-baseline > gpurun_out/r3f_bf16.json 2> gpurun_out/r3f_bf16.err
timeout -k 10 300 python3 bench.py --mode edge-shard --no-cpu-baseline > gpurun_out/r3f_edge.json 2> gpurun_out/r3f_edge.err
echo "lines done"
bash scratch/kstats.sh r3f_c2_4096k python3 bench.py --config c2 --molecules 4096 --steps 5 --warmup 2 --no-precondition > /dev/null
echo "all done"
