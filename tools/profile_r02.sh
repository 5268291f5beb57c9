#!/bin/bash
# Usage (GPU box, repo root): bash tools/profile_r02.sh
# Everything the profiles/r02_* files come from: bench lines, rocprofv3 kernel traces and counter passes for the
# 4096 x 30 benchmark and for 256 x 1024 (config 4).  Every GPU step has its own time limit; the script stops at
# the first failure.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
b() { timeout -k 10 300 python3 bench.py "$@"; }
b > $O/bench_default.json 2> $O/bench_default.err
b --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2>> $O/bench_default.err
: > $O/bench_variants.jsonl
for args in "--regime packed" "--periodic" "--keep-indices sensed" "--keep-indices both" "--envs-per-gpu 8192" \
            "--envs-per-gpu 256 --agents 1024 --streams 1" "--envs-per-gpu 256 --agents 1024 --streams 2"; do
  line=$(b --no-cpu-baseline $args 2>> $O/bench_variants.err)
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); d={'bench_args': sys.argv[2], **d}; print(json.dumps(d))" "$line" "--no-cpu-baseline $args" >> $O/bench_variants.jsonl
done
line=$(MARL_AMD_NO_LATTICE=1 b --no-cpu-baseline 2>> $O/bench_variants.err)
python3 -c "import json,sys; d=json.loads(sys.argv[1]); d={'bench_args': sys.argv[2], **d}; print(json.dumps(d))" "$line" "MARL_AMD_NO_LATTICE=1 --no-cpu-baseline (every cell tested against every agent)" >> $O/bench_variants.jsonl
for v in 20 36; do                                       # 256 x 1024 without cell lists (+16) / as three launches (+32)
  line=$(MARL_AMD_FAST_T=$v b --no-cpu-baseline --envs-per-gpu 256 --agents 1024 --streams 2 2>> $O/bench_variants.err)
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); d={'bench_args': sys.argv[2], **d}; print(json.dumps(d))" "$line" "MARL_AMD_FAST_T=$v --no-cpu-baseline --envs-per-gpu 256 --agents 1024 --streams 2" >> $O/bench_variants.jsonl
done
echo bench-done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 $R/bench.py --no-cpu-baseline > $O/trace_default.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --no-cpu-baseline --envs-per-gpu 256 --agents 1024 --streams 1 > $O/trace_c4.log 2>&1
echo trace-done
cd $R
bash tools/pmc.sh r02/pmc_default --steps 50
python3 tools/pmc_summary.py $O/pmc_default k_step --json $O/pmc_default.json > $O/pmc_default.txt
bash tools/pmc.sh r02/pmc_c4 --steps 50 --envs-per-gpu 256 --agents 1024 --streams 1
python3 tools/pmc_summary.py $O/pmc_c4 "k_step<" --json $O/pmc_c4.json --envs 256 --agents 1024 --streams 1 > $O/pmc_c4.txt
echo profile-done
