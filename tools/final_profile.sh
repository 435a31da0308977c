#!/bin/bash
# Round-end evidence for one bench workload: bench line, rocprofv3 kernel trace of the same command, PMC passes of one launch.
# Run on the GPU box from the repo root: bash tools/final_profile.sh <tag> [workload] [pmc spp] [pmc size]
TAG=${1:-final}; WL=${2:-cornell}; SPP=${3:-256}; SIZE=${4:-}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout -k 10 500 python -u bench.py --workload $WL > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { echo "bench failed"; tail -5 gpurun_out/bench_$TAG.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/${TAG}_kt -o b -- python3 bench.py --workload $WL --no-cpu-baseline > gpurun_out/bench_${TAG}_prof.json 2> gpurun_out/bench_${TAG}_prof.err || { echo "profiled bench failed"; exit 1; }
WORLD=$WL MIRT_SIZE=$SIZE bash tools/pmc.sh $SPP ${TAG}_pmc > gpurun_out/pmc_$TAG.log 2>&1 || { echo "pmc failed"; tail -5 gpurun_out/pmc_$TAG.log; exit 1; }
tail -1 gpurun_out/bench_$TAG.json | cut -c1-300
