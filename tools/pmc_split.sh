cd /tmp && export TMPDIR=/tmp
export MARL_AMD_BENCH_SETTLE=50
out=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc_split; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pass1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 --repeats 1 --envs-per-gpu 256 --agents 1024 --streams 1 > $out/pass1.log 2>&1 || { tail -3 $out/pass1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $out/pass2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 --repeats 1 --envs-per-gpu 256 --agents 1024 --streams 1 > $out/pass2.log 2>&1 || { tail -3 $out/pass2.log; exit 1; }
for k in k_observe_part "k_step<" k_swarm_stats; do echo "== $k"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "$k"; done
rm -rf $out
