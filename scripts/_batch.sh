cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/b13
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b13/trace -o t -- \
    python bench.py --steps 12 --warmup 3 --no-side-configs --no-cpu-baseline --no-strict-range --no-roofline --no-partition-check > /dev/null 2>&1 )
python scripts/trace_step_positions.py gpurun_out/b13/trace > gpurun_out/b13/step_positions.log 2>&1
rm -rf gpurun_out/b13/trace
