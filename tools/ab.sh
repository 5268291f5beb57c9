#!/bin/bash
# Usage (GPU box, repo root): bash tools/ab.sh <libA.so> <libB.so>   (paths relative to marl_amd/csrc/build/)
# Same-box A/B of two builds of the library: step time of the default bench, alternating, and VALU counts.
A=$GRAFT_REPO_ROOT/marl_amd/csrc/build/$1; B=$GRAFT_REPO_ROOT/marl_amd/csrc/build/$2
for k in 1 2 3; do
  for lib in $A $B; do
    echo -n "$(basename $lib) "
    MARL_AMD_LIB=$lib timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1500 --warmup 400 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
  done
done
for lib in $A $B; do echo -n "$(basename $lib) "; bash tools/pmc_ab.sh ab_$(basename $lib .so) $lib | grep INSTS_VALU; done
