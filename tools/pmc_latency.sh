#!/bin/bash
# Usage (GPU box, repo root): bash tools/pmc_latency.sh <name> [bench args...]
# What the waves of the step kernel wait for: three counter passes (eight SQ counters each) over bench.py --steps 50 --
# wait cycles, instruction-fetch, vector/scalar-memory and LDS occupancy levels (level / instructions = mean latency in
# cycles), cycles each instruction class is active, queue-full stalls.  Summary on stdout; raw CSVs removed.
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/lat_$name
cd /tmp && export TMPDIR=/tmp
p=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM" \
            "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  p=$((p + 1)); mkdir -p $out/pass$p
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 50 "$@" > $out/pass$p.log 2>&1 || { echo "pass $p failed"; tail -3 $out/pass$p.log; exit 1; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out k_step
rm -rf $out
