#!/bin/bash
# Usage (GPU box, repo root; after `make -C marl_amd/csrc exits`): bash tools/phase_valu.sh
# Vector/scalar/LDS instruction counts of k_step_fast up to each phase boundary (the accounting build returns
# there), one counter pass per boundary; differences between consecutive lines are the phases' own counts.
# 13 = the whole kernel without the k-NN block.
lib=$GRAFT_REPO_ROOT/marl_amd/csrc/build/libAssemblyEnv_exits.so
for x in 1 2 3 4 5 6 0 13; do
  export MARL_AMD_DBG_EXIT=$x
  echo "== exit $x"
  bash $GRAFT_REPO_ROOT/tools/pmc_ab.sh exits_$x $lib | grep -E "INSTS_VALU|INSTS_SALU|INSTS_LDS|pass failed" || exit 1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/exits_$x            # the raw per-launch CSVs are large
done
