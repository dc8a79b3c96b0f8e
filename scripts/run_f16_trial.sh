mkdir -p gpurun_out/f16
V=$PWD/graphs4cfd_amd/lib_variants
( echo "--- bx6i check HEAD"; timeout 300 python scripts/bx6i_check.py --time 2>&1 | tail -3
  echo "--- bx6i check late W"; G4C_LIB_PATH=$V/libg4c_lw.so timeout 300 python scripts/bx6i_check.py --time 2>&1 | tail -3
  echo "--- stamps late W"; timeout 300 python scripts/bx6i_stamps.py $V/libg4c_lwt.so 2>&1 | tail -19
  echo "--- bench HEAD"; timeout 600 python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['largest_launch']['avg_launch_us'], d['roofline']['avg_launch_us'])"
  echo "--- bench late W"; G4C_LIB_PATH=$V/libg4c_lw.so timeout 600 python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['largest_launch']['avg_launch_us'], d['roofline']['avg_launch_us'])"
  echo "--- bench late W, bx6i from 100k rows"; G4C_BX6I_MIN_ROWS=100000 G4C_LIB_PATH=$V/libg4c_lw.so timeout 600 python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['largest_launch']['avg_launch_us'], d['roofline']['avg_launch_us'])"
  echo "--- smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
) > gpurun_out/f16/run7.log 2>&1
cat gpurun_out/f16/run7.log
