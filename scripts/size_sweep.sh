#!/bin/bash
# A rank's share of a mesh as a single-GPU rollout (before halo rows and exchanges): the by-construction ceiling of the partitioned run.
# Usage (GPU box): bash scripts/size_sweep.sh > gpurun_out/rNN_size_sweep.log
cd "$GRAFT_REPO_ROOT"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
echo "# a rank's share of the 100k-node headline mesh as a single-GPU rollout (bench.py --nodes N, 3-scale MuS-GNN, f16x3, hipGraph replay): nodes steps/s ms/step"
for n in 100000 50000 25000 12500; do
  timeout 600 python bench.py --nodes $n --steps 200 --no-cpu-baseline --no-roofline --no-side-configs --no-strict-range 2>/dev/null | tail -1 | line $n
done
echo "# config 5 (4-scale MuS-GNN, 3-D mesh): the 1M-node mesh on one GPU and a rank's share at 2 / 4 / 8 GPUs (bench.py --workload c5-1gpu --nodes N): nodes steps/s ms/step"
for n in 1000000 500000 250000 125000; do
  timeout 900 python bench.py --workload c5-1gpu --nodes $n --steps 30 --no-cpu-baseline --no-roofline --no-side-configs --no-strict-range 2>/dev/null | tail -1 | line $n
done
