#!/bin/bash
# Usage (GPU box, repo root): bash tools/c4_ab.sh
# 256 x 1024 (config 4): pair passes over all agents (schedule + 16) against cell lists at several bin sizes
# (MARL_AMD_BIN_REACH = sensing radius / bin side), one and two launches in flight.
for rep in 1 2; do
  for v in "MARL_AMD_FAST_T=20" "MARL_AMD_BIN_REACH=2" "MARL_AMD_BIN_REACH=3" "MARL_AMD_BIN_REACH=4"; do
    for s in 1 2; do
      echo "== $v streams $s"
      env $v timeout -k 10 150 python bench.py --no-cpu-baseline --envs-per-gpu 256 --agents 1024 --streams $s --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step']*1e3, 'us', d['value'], d['roofline']['frac'])" || exit 1
    done
  done
done
