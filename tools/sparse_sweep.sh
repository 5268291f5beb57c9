cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in libAssemblyEnv libSP12 libSP40 libSP64; do
  for reg in random packed; do
    echo -n "$lib $reg "; MARL_AMD_LIB=$GRAFT_REPO_ROOT/marl_amd/csrc/build/$lib.so timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 400 --regime $reg 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'
  done
done
done
