#!/bin/bash
# Usage (GPU box, repo root; after `make -C marl_amd/csrc exits`): bash tools/phase_c4.sh
# Instruction and cycle counts of k_step<float> on 256 x 1024 up to each phase boundary (the accounting build returns
# there); differences between consecutive lines are the phases' own counts.  Exits: 1 load, 2 dynamics + state write,
# 3 k-NN + agent block, 4 cell pass, 5 list sizes + target block, 6 emission, 7 reward, 0 whole kernel (+ statistics).
lib=$GRAFT_REPO_ROOT/marl_amd/csrc/build/libAssemblyEnv_exits.so
export MARL_AMD_LIB=$lib
cd /tmp && export TMPDIR=/tmp
for x in 1 2 3 4 5 6 7 0; do
  export MARL_AMD_DBG_EXIT=$x
  out=$GRAFT_REPO_ROOT/gpurun_out/c4_exit_$x; mkdir -p $out
  echo "== exit $x"
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pass1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 --repeats 1 --envs-per-gpu 256 --agents 1024 --streams 1 "$@" > $out/pass1.log 2>&1 || { echo "pass failed"; tail -3 $out/pass1.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "k_step<"
  rm -rf $out                                            # the raw per-launch CSVs are large
done
