#!/bin/bash
# Usage (GPU box, repo root): bash tools/profile_r03.sh [part ...]   (parts: bench trace pmc phases; default all)
# Everything the profiles/r03_* bench, trace and counter files come from: bench lines, rocprofv3 kernel traces and
# counter passes for the 4096 x 30 benchmark and for 256 x 1024, per-phase instruction counts and cycles of the final
# binaries.  Every GPU step has its own time limit; the script stops at the first failure.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
parts=${@:-bench trace pmc phases}
cd $R
b() { timeout -k 10 300 python3 bench.py "$@"; }
tag() { python3 -c "import json,sys; d=json.loads(sys.argv[1]); d={'bench_args': sys.argv[2], **d}; print(json.dumps(d))" "$1" "$2"; }
for part in $parts; do
case $part in
bench)
  b > $O/bench_default.json 2> $O/bench_default.err
  b --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2>> $O/bench_default.err
  : > $O/bench_variants.jsonl
  for args in "--regime packed" "--periodic" "--keep-indices sensed" "--keep-indices both" "--envs-per-gpu 8192" "--streams 2" "--streams 4" \
              "--envs-per-gpu 256 --agents 1024 --streams 1" "--envs-per-gpu 256 --agents 1024 --streams 2" \
              "--envs-per-gpu 256 --agents 1024 --streams 4" "--envs-per-gpu 256 --agents 1024 --steps 20 --warmup 5"; do
    line=$(b --no-cpu-baseline $args 2>> $O/bench_variants.err)
    tag "$line" "--no-cpu-baseline $args" >> $O/bench_variants.jsonl
  done
  line=$(MARL_AMD_NO_LATTICE=1 b --no-cpu-baseline 2>> $O/bench_variants.err)
  tag "$line" "MARL_AMD_NO_LATTICE=1 --no-cpu-baseline (every cell tested against every agent)" >> $O/bench_variants.jsonl
  line=$(MARL_AMD_NO_SWARM_KERNEL=1 b --no-cpu-baseline --envs-per-gpu 256 --agents 1024 2>> $O/bench_variants.err)
  tag "$line" "MARL_AMD_NO_SWARM_KERNEL=1 --no-cpu-baseline --envs-per-gpu 256 --agents 1024 (the generic instantiation of the step kernel)" >> $O/bench_variants.jsonl
  echo bench-done ;;
trace)
  cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 $R/bench.py --no-cpu-baseline > $O/trace_default.log 2>&1
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --no-cpu-baseline --envs-per-gpu 256 --agents 1024 > $O/trace_c4.log 2>&1
  cd $R
  cp $(ls $O/trace_default/*/*kernel_stats.csv | head -1) $O/bench_default_kernel_stats.csv
  cp $(ls $O/trace_c4/*/*kernel_stats.csv | head -1) $O/c4_kernel_stats.csv
  python3 tools/kernel_stats_summary.py $O/bench_default_kernel_stats.csv 4096 30 3 $O/bench_default_kernel_trace.json $O/trace_default.log
  python3 tools/kernel_stats_summary.py $O/c4_kernel_stats.csv 256 1024 4 $O/c4_kernel_trace.json $O/trace_c4.log
  rm -rf $O/trace_default $O/trace_c4
  echo trace-done ;;
pmc)
  bash tools/pmc.sh r03p/pmc_default --steps 50
  python3 tools/pmc_summary.py $O/pmc_default k_step --json $O/pmc.json --streams 3 > $O/pmc_k_step_fast_4096x30.txt
  rm -rf $O/pmc_default
  bash tools/pmc.sh r03p/pmc_c4 --steps 50 --envs-per-gpu 256 --agents 1024 --streams 1
  python3 tools/pmc_summary.py $O/pmc_c4 "k_step<" --json $O/c4_pmc.json --envs 256 --agents 1024 --streams 1 > $O/c4_pmc_k_step_256x1024.txt
  rm -rf $O/pmc_c4
  echo pmc-done ;;
phases)
  bash tools/phase_valu.sh > $O/phase_valu_raw.txt 2>&1
  python3 tools/phase_table.py $O/phase_valu_raw.txt "P0 load,P1 + P2 dynamics and state write,P3 k-NN + lattice rows + nearest cell,(P4: inside P3 on the lattice path),P5 slot lists,P6 emission,P7 reward + statistics,the k-NN block alone with the sign flipped (this run is the whole kernel without it)" > $O/phase_instruction_counts.txt
  bash tools/phase_c4.sh > $O/c4_phase_raw.txt 2>&1
  python3 tools/phase_table.py $O/c4_phase_raw.txt "load,dynamics + state write,k-NN + agent block,cell pass,list sizes + target block,emission,reward,statistics" > $O/c4_phase_counts.txt
  timeout -k 10 120 python3 tools/diag_phases.py 1366 30 3 > $O/phase_cycles.txt 2>&1
  timeout -k 10 120 python3 tools/diag_phases.py 256 1024 > $O/c4_phase_cycles.txt 2>&1
  echo phases-done ;;
esac
done
echo profile-done
