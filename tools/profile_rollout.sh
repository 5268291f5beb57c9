#!/bin/bash
# Usage (GPU box, repo root): bash tools/profile_rollout.sh
# Config 5 lines (tools/bench_rollout.py: float32 / bf16 through PyTorch / bf16 through the library's policy kernel,
# with and without the graphed train phase, and as two shards on streams of their own), the policy kernel alone (tools/bench_policy.py), and a rocprofv3 kernel
# trace of one fused-bf16 run.  Outputs under gpurun_out/r02/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
: > $O/rollout.jsonl
for f in "" "--act-dtype bfloat16 --no-fused" "--act-dtype bfloat16" "--act-dtype bfloat16 --graph" "--act-dtype bfloat16 --graph --shards 2"; do
  timeout -k 10 300 python3 tools/bench_rollout.py --episodes 4 $f 2>> $O/rollout.err >> $O/rollout.jsonl
done
: > $O/policy.jsonl
for a in "4096 30 256" "4096 30 128" "256 1024 256" "2048 30 256"; do
  timeout -k 10 120 python3 tools/bench_policy.py $a 2>> $O/rollout.err >> $O/policy.jsonl
done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_rollout -- python3 $R/tools/bench_rollout.py --episodes 2 --act-dtype bfloat16 > $O/trace_rollout.log 2>&1
echo rollout-profile-done
