#!/bin/bash
# The training-path part of refresh_artifacts.sh alone (same files), for when only autograd.py / train_ops.hip changed.
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT"
A=gpurun_out/artifacts_$TAG; mkdir -p $A
timeout -k 10 900 python scripts/bench_train.py --steps 10 --cpu-steps 1 --phases 2> $A/train_stderr.log | tail -1 > $A/${TAG}_train_bench_100k.json
timeout -k 10 300 python scripts/bench_train.py --nodes 10000 --model NsTwoScaleGNN --steps 30 --cpu-steps 1 2>/dev/null | tail -1 > $A/${TAG}_train_bench_10k.json
rm -rf $A/prof_train
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $A/prof_train -o t -- \
    python scripts/bench_train.py --steps 3 --cpu-steps 0 > $A/prof_train_stdout.log 2> $A/prof_train_stderr.log )
cp $(find $A/prof_train -name '*kernel_stats.csv' | head -1) $A/${TAG}_train_rocprofv3_kernel_stats.csv 2>/dev/null
rm -rf $A/prof_train
for v in "" "--no-host-copies"; do echo "host copies of index tensors: ${v:-kept}"; timeout -k 5 300 python scripts/bench_fit_batches.py $v 2>&1 | grep -i "fresh\|same"; done > $A/${TAG}_fit_fresh_batches.log
timeout -k 5 300 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $A/${TAG}_pytest_gpu.log
ls -la $A
