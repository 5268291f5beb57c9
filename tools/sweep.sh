#!/bin/bash
# Usage (GPU box, repo root): bash tools/sweep.sh <out-file-under-gpurun_out>
# Step time of the batched path over environment counts and k_step_fast schedules (MARL_AMD_FAST_T).
out=$GRAFT_REPO_ROOT/gpurun_out/$1
: > $out
for E in 2048 2560 4096 8192 16384 32768; do
  for T in 1 2 4; do
    r=$(MARL_AMD_FAST_T=$T timeout -k 10 120 python3 bench.py --envs-per-gpu $E --steps 200 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.e+]*, \|"ms_per_step": [0-9.]*') || { echo "E=$E T=$T failed" >> $out; exit 1; }
    echo "E=$E T=$T $r" | tr '\n' ' ' >> $out; echo >> $out
  done
done
echo sweep-done
