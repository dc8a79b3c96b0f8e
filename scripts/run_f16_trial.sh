mkdir -p gpurun_out/f16
V=$PWD/graphs4cfd_amd/lib_variants
( echo "--- bx6i check f16x3"; timeout 300 python scripts/bx6i_check.py --time 2>&1 | tail -3
  echo "--- stamps"; timeout 300 python scripts/bx6i_stamps.py $V/libg4c_timing.so 2>&1 | tail -19
  echo "--- bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['largest_launch']['avg_launch_us'], d['roofline']['avg_launch_us'])"
  echo "--- bench c2"; timeout 600 python bench.py --workload c2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
) > gpurun_out/f16/run5.log 2>&1
cat gpurun_out/f16/run5.log
