#!/bin/bash
# Round artifacts, produced on the GPU box into gpurun_out/artifacts_<tag>/ (copy them into profiles/ afterwards):
#   pytest -m gpu tail, default bench line, rocprofv3 kernel stats of the bench, PMC HBM traffic, MFMA ceiling, side configs.
# Usage: gpurun --timeout 2400 -- 'bash scripts/refresh_artifacts.sh r01'
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT"
A=gpurun_out/artifacts_$TAG; rm -rf $A; mkdir -p $A
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $A/${TAG}_pytest_gpu.log
timeout 900 python bench.py > $A/bench_stdout.log 2> $A/bench_stderr.log; tail -1 $A/bench_stdout.log > $A/${TAG}_bench_n1.json
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $A/prof -o p -- \
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $A/prof_stdout.log 2> $A/prof_stderr.log )
tail -1 $A/prof_stdout.log > $A/${TAG}_bench_under_rocprofv3.json
cp $(find $A/prof -name '*kernel_stats.csv' | head -1) $A/${TAG}_rocprofv3_kernel_stats.csv
timeout 900 bash scripts/collect_pmc_traffic.sh $TAG > $A/pmc.log 2>&1; cp profiles/${TAG}_pmc_traffic.json $A/ 2>/dev/null
hipcc -w --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/micro/mfma_peak.hip && /tmp/mfma_peak 2000 > $A/${TAG}_mfma_peak.log 2>&1
{
  echo "# side configurations (bench.py --no-cpu-baseline --no-roofline), one JSON line each"
  timeout 300 python bench.py --nodes 12500 --steps 50 --no-cpu-baseline --no-roofline | tail -1
  timeout 300 python bench.py --model NsTwoScaleGNN --nodes 10000 --steps 50 --no-cpu-baseline --no-roofline | tail -1
  timeout 300 python bench.py --model NsFourScaleGNN --nodes 100000 --steps 20 --no-cpu-baseline --no-roofline | tail -1
  timeout 300 python scripts/bench_remus.py 2>&1 | tail -3
  timeout 300 python scripts/bench_mugs.py 2>&1 | tail -2
  timeout 300 python bench.py --precision fp32 --steps 20 --no-cpu-baseline | tail -1
} > $A/${TAG}_side_configs.log 2>&1
# training path: step time + per-phase HIP-event times + the CPU leg; then the per-kernel view of the same step
timeout -k 10 900 python scripts/bench_train.py --steps 10 --cpu-steps 1 --phases 2> $A/train_stderr.log | tail -1 > $A/${TAG}_train_bench_100k.json
timeout -k 10 300 python scripts/bench_train.py --nodes 10000 --model NsTwoScaleGNN --steps 30 --cpu-steps 1 2>/dev/null | tail -1 > $A/${TAG}_train_bench_10k.json
rm -rf $A/prof
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $A/prof_train -o t -- \
    python scripts/bench_train.py --steps 3 --cpu-steps 0 > $A/prof_train_stdout.log 2> $A/prof_train_stderr.log )
cp $(find $A/prof_train -name '*kernel_stats.csv' | head -1) $A/${TAG}_train_rocprofv3_kernel_stats.csv 2>/dev/null
rm -rf $A/prof_train; ls -la $A
