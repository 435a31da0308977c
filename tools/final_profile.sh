#!/bin/bash
# Round-end evidence: bench line, rocprofv3 kernel trace of the same command, PMC passes of one 256-spp launch.
# Run on the GPU box from the repo root: bash tools/final_profile.sh <tag>
TAG=${1:-final}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout -k 10 400 python -u bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { echo "bench failed"; tail -5 gpurun_out/bench_$TAG.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/${TAG}_kt -o b -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_${TAG}_prof.json 2> gpurun_out/bench_${TAG}_prof.err || { echo "profiled bench failed"; exit 1; }
bash tools/pmc.sh 256 ${TAG}_pmc > gpurun_out/pmc_$TAG.log 2>&1 || { echo "pmc failed"; exit 1; }
tail -1 gpurun_out/bench_$TAG.json | cut -c1-300
