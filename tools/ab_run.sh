#!/bin/bash
# A/B timing of two builds of libmirt on the GPU box: bash tools/ab_run.sh <libA.so> <libB.so> [rounds]
# (library names under raytracinginaweekend_amd/lib; build the baseline from another checkout with the flags of build.py).
# Each round renders the four quickbench workloads with either library in turn; MIRT_OPTS passes options to both.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
L=raytracinginaweekend_amd/lib
A=${1:-libmirt_base.so}; B=${2:-libmirt.so}; N=${3:-2}
for i in $(seq $N); do
  for lib in $A $B; do
    echo "== $lib"
    MIRT_LIB=$L/$lib timeout -k 10 100 python -u tools/quickbench.py 512 cornell 3 | tail -1 || exit 1
    MIRT_LIB=$L/$lib timeout -k 10 100 python -u tools/quickbench.py 100 spheres4 3 | tail -1 || exit 1
    MIRT_LIB=$L/$lib timeout -k 10 100 python -u tools/quickbench.py 128 spheres 3 | tail -1 || exit 1
    MIRT_SIZE=1920x1080 MIRT_LIB=$L/$lib timeout -k 10 100 python -u tools/quickbench.py 32 spheres1m 3 | tail -1 || exit 1
  done
done
