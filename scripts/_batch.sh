cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/b7
timeout 300 python scripts/_dbg_fuse.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/b7/dbg.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | cut -c1-250 > gpurun_out/b7/pytest_gpu_full.log
tail -c 5000 gpurun_out/b7/pytest_gpu_full.log > gpurun_out/b7/pytest_tail.log
