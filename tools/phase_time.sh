#!/bin/bash
# Usage (GPU box, repo root; after `make -C marl_amd/csrc exits`): bash tools/phase_time.sh [bench args...]
# Step time of the benchmark with the step kernel returning at each phase boundary in turn (accounting build,
# MARL_AMD_DBG_EXIT): differences between consecutive lines are what each phase adds to the step time with every
# launch of the run in flight, i.e. its cost in throughput rather than its share of one workgroup's lifetime.
# (A kernel that returns early leaves the state as it was: the agents stay where the reset put them.)
lib=$GRAFT_REPO_ROOT/marl_amd/csrc/build/libAssemblyEnv_exits.so
for x in 1 2 3 4 5 6 0; do
  echo -n "exit $x: "
  MARL_AMD_DBG_EXIT=$x MARL_AMD_LIB=$lib timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 200 "$@" 2>/dev/null | grep -o 'ms_per_step": [0-9.]*' || exit 1
done
