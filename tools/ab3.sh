#!/bin/bash
# Usage (GPU box, repo root): bash tools/ab3.sh <lib1.so> <lib2.so> ...   (paths relative to marl_amd/csrc/build/)
# Same-box comparison of several builds of the library: step time of the default bench, alternating, and VALU counts.
for k in 1 2 3; do
  for n in "$@"; do
    lib=$GRAFT_REPO_ROOT/marl_amd/csrc/build/$n
    echo -n "$n "
    MARL_AMD_LIB=$lib timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1500 --warmup 400 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
  done
done
for n in "$@"; do lib=$GRAFT_REPO_ROOT/marl_amd/csrc/build/$n; echo -n "$n "; bash tools/pmc_ab.sh ab_$(basename $n .so) $lib | grep INSTS_VALU; done
