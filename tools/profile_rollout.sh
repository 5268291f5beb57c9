#!/bin/bash
# Usage (GPU box, repo root): bash tools/profile_rollout.sh
# Config 5 lines (tools/bench_rollout.py).  The reference's precision first: the float32 policy through PyTorch with the
# agent rows made by PyTorch's strided copy (as before), by the library's transpose kernel, with PyTorch's GEMM tuning, as
# two shards on streams of their own, and with the train phase replayed from HIP graphs.  Then, labelled as such in every
# line, the lower-precision (bf16) rollouts.  Outputs under gpurun_out/r03/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
: > $O/rollout.jsonl
for f in "--strided-copy" "" "--tunableop" "--tunableop --shards 2" "--tunableop --graph" "--tunableop --shards 2 --graph" \
         "--act-dtype bfloat16 --no-fused" "--act-dtype bfloat16 --graph" "--act-dtype bfloat16 --graph --shards 2"; do
  line=$(timeout -k 10 300 python3 tools/bench_rollout.py --episodes 4 $f 2>> $O/rollout.err)
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); d={'bench_args': sys.argv[2], **d}; print(json.dumps(d))" "$line" "--episodes 4 $f" >> $O/rollout.jsonl
done
echo rollout-profile-done
