#!/bin/bash
# Usage (GPU box, repo root): bash tools/pmc_ab.sh <outdir-under-gpurun_out> <library.so> [bench args...]
# Instruction-mix counters of one build of the library (one rocprofv3 --pmc pass).
out=$GRAFT_REPO_ROOT/gpurun_out/$1; lib=$2; shift; shift
mkdir -p $out
export MARL_AMD_LIB=$lib
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/pass1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 50 "$@" > $out/pass1.log 2>&1 || { echo "pass failed"; tail -3 $out/pass1.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out k_step
