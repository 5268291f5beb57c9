#!/bin/bash
# Usage (GPU box, repo root): bash tools/ab_c4.sh <lib1.so> <lib2.so> ...   (paths relative to marl_amd/csrc/build/)
# Same-box comparison of several builds of the library on the large swarm (256 x 1024, the bench's default streams), alternating.
for k in 1 2 3; do
  for n in "$@"; do
    lib=$GRAFT_REPO_ROOT/marl_amd/csrc/build/$n
    echo -n "$n "
    MARL_AMD_LIB=$lib timeout -k 10 150 python3 bench.py --no-cpu-baseline --envs-per-gpu 256 --agents 1024 --steps 400 --warmup 50 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
  done
done
