# A/B of build variants of the dual-tile kernel in f16x3 mode, then the GPU test suite with f16x3 as the default arithmetic
mkdir -p gpurun_out/f16
( for v in "" _w3 _v12 _v5 _w3v12; do
    echo "--- bx6i check f16x3, lib libg4c$v.so"; G4C_LIB_PATH=$PWD/graphs4cfd_amd/lib/libg4c$v.so timeout 300 python scripts/bx6i_check.py --time 2>&1 | tail -3
  done
  for mr in 400000 100000 20000; do
    echo "--- bench f16x3 G4C_BX6I_MIN_ROWS=$mr"; G4C_BX6I_MIN_ROWS=$mr timeout 600 python bench.py --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['largest_launch']['avg_launch_us'])"
  done
  echo "--- bench f16x3 w3"; G4C_LIB_PATH=$PWD/graphs4cfd_amd/lib/libg4c_w3.so timeout 600 python bench.py --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['largest_launch']['avg_launch_us'])"
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
) > gpurun_out/f16/run2.log 2>&1
tail -60 gpurun_out/f16/run2.log
