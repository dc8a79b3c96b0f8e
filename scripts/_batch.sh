cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/b11
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "node_kernel or fused_mp_layer or range" 2>&1 | tail -5 > gpurun_out/b11/pytest_subset.log
timeout 600 python scripts/mp_layer_check.py --time --no-check 2>&1 | grep -v "amdgpu.ids" | grep -v "^ok" > gpurun_out/b11/mp_layer_time.log
