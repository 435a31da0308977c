#!/bin/bash
# A/B of library builds on one box: bash tools/ab_libs.sh <spp> <world> <lib> [<lib> ...]   (names under raytracinginaweekend_amd/lib/)
# Three interleaved rounds of tools/quickbench.py (device ms of the last of three renders each).
L=raytracinginaweekend_amd/lib
spp=$1; world=$2; shift 2
for i in 1 2 3; do for lib in "$@"; do echo -n "$lib  "; MIRT_LIB=$L/$lib python tools/quickbench.py $spp $world 3 | tail -1; done; done
