#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
# One rocprofv3 --pmc pass per counter group (separate runs), kernel-trace not combined with counters.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" \
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH" \
 "FETCH_SIZE" \
 "WRITE_SIZE" \
 "GRBM_GUI_ACTIVE GRBM_COUNT" \
 "TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TOTAL_WRITE TCP_PENDING_STALL_CYCLES" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/pass$i.log; exit 1; }
done
echo pmc-done
