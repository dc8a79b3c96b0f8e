cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/b17
{ echo "# a rank's share of the 100k-node headline mesh as a single-GPU rollout (bench.py --nodes N, 3-scale MuS-GNN, f16x3, hipGraph replay): nodes steps/s ms/step"
for n in 100000 50000 25000 12500; do
  timeout 600 python bench.py --nodes $n --steps 100 --no-cpu-baseline --no-roofline --no-side-configs --no-strict-range 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print($n, round(d['value'],1), round(d['ms_per_step'],3))"
done; } > gpurun_out/b17/size_sweep.log 2>&1
