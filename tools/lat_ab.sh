#!/bin/bash
# Usage (GPU box, repo root; after `make -C marl_amd/csrc all diag`): bash tools/lat_ab.sh
# Lattice path against the all-cells path of the same build: step time (alternating runs), instruction counters,
# phase cycles of a workgroup's first wave.
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 -u -m pytest tests/test_gpu_lattice.py -x -q --timeout 120 --timeout-method thread
for k in 1 2; do
  echo -n "lattice "; timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 400 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
  echo -n "allcells "; MARL_AMD_NO_LATTICE=1 timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 400 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
done
echo -n "lattice packed "; timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 400 --regime packed 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
echo -n "allcells packed "; MARL_AMD_NO_LATTICE=1 timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 400 --regime packed 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
echo "== counters lattice"; bash tools/pmc_ab.sh ab_lat $GRAFT_REPO_ROOT/marl_amd/csrc/build/libAssemblyEnv.so
echo "== diag lattice"; timeout -k 10 120 python3 tools/diag_phases.py 2048 2>/dev/null
