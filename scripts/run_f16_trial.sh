mkdir -p gpurun_out/f16
run() { env "$@" timeout 300 python bench.py $WL --no-cpu-baseline --no-roofline --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))"; }
( for WL in "--workload headline" "--model NsFourScaleGNN --nodes 100000"; do
    echo "=== $WL"
    for rep in 1 2 3; do
      echo -n "default: "; run A=1
      echo -n "FUSE 20000: "; run G4C_FUSE_AGG_MIN_ROWS=20000
      echo -n "FUSE 20000 + AOL 20000: "; run G4C_FUSE_AGG_MIN_ROWS=20000 G4C_AGG_ON_LOAD_MIN_ROWS=20000
      echo -n "FUSE 20000 + AOL 20000 + HOIST 50000: "; run G4C_FUSE_AGG_MIN_ROWS=20000 G4C_AGG_ON_LOAD_MIN_ROWS=20000 G4C_HOIST_MIN_ROWS=50000
    done
  done
) > gpurun_out/f16/run10.log 2>&1
cat gpurun_out/f16/run10.log
