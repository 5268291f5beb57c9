#!/bin/bash
# Usage (GPU box, repo root): bash tools/clock_probe.sh <lib.so under marl_amd/csrc/build> [bench args]
# Samples the GPU's clock and power (rocm-smi) while the default bench runs on the given build of the library.
lib=$GRAFT_REPO_ROOT/marl_amd/csrc/build/$1; shift
MARL_AMD_LIB=$lib timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 60000 --warmup 400 --repeats 5 "$@" > /tmp/clock_probe_bench.json 2>/dev/null &
pid=$!
sleep 6
for k in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done
wait $pid
grep -o 'ms_per_step": [0-9.]*' /tmp/clock_probe_bench.json
