mkdir -p gpurun_out/f16
( echo "--- stamps (f16x3, 3 workgroups per CU)"; timeout 300 python scripts/bx6i_stamps.py graphs4cfd_amd/lib/libg4c_bx6i_timing.so 2>&1 | tail -24
  timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
  echo "--- bench"; timeout 600 python bench.py 2>&1 | tail -1
) > gpurun_out/f16/run3.log 2>&1
tail -60 gpurun_out/f16/run3.log
