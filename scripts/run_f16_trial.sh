mkdir -p gpurun_out/f16
run() { env "$@" timeout 300 python bench.py $WL --no-cpu-baseline --no-roofline --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['outputs_finite'])"; }
( for WL in "--workload headline" "--nodes 50000"; do
    echo "=== $WL"
    for rep in 1 2; do
      echo -n "default: "; run A=1
      echo -n "RT2 from 50000 rows: "; run G4C_BX6_RT2_ROWS=50000
      echo -n "RT2 from 200000 rows: "; run G4C_BX6_RT2_ROWS=200000
    done
  done
  G4C_BX6_RT2_ROWS=50000 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline or mus_models or heads" 2>&1 | tail -3
) > gpurun_out/f16/run12.log 2>&1
cat gpurun_out/f16/run12.log
