set -e
cd $GRAFT_REPO_ROOT
for k in 1 2; do
  echo -n "lattice "; timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 400 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
  echo -n "allcells "; MARL_AMD_NO_LATTICE=1 timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 400 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
done
echo "== counters lattice"; bash tools/pmc_ab.sh ab_lat $GRAFT_REPO_ROOT/marl_amd/csrc/build/libAssemblyEnv.so
echo "== counters allcells"; MARL_AMD_NO_LATTICE=1 bash tools/pmc_ab.sh ab_nolat $GRAFT_REPO_ROOT/marl_amd/csrc/build/libAssemblyEnv.so
echo "== diag lattice"; timeout -k 10 120 python3 tools/diag_phases.py 2048
echo "== diag allcells"; MARL_AMD_NO_LATTICE=1 timeout -k 10 120 python3 tools/diag_phases.py 2048
