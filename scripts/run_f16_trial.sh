mkdir -p gpurun_out/f16
V=$PWD/graphs4cfd_amd/lib_variants
( echo "--- bx6i check f16x3 HEAD"; timeout 300 python scripts/bx6i_check.py --time 2>&1 | tail -3
  echo "--- bx6i check f16x3 row stores"; G4C_LIB_PATH=$V/libg4c_rs.so timeout 300 python scripts/bx6i_check.py --time 2>&1 | tail -3
  echo "--- bench HEAD"; timeout 600 python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['largest_launch']['avg_launch_us'], d['roofline']['avg_launch_us'])"
  echo "--- bench row stores"; G4C_LIB_PATH=$V/libg4c_rs.so timeout 600 python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['largest_launch']['avg_launch_us'], d['roofline']['avg_launch_us'])"
) > gpurun_out/f16/run6.log 2>&1
cat gpurun_out/f16/run6.log
