run() { env "$@" timeout 300 python bench.py $WL --no-cpu-baseline --no-roofline --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['outputs_finite'])"; }
for WL in "--nodes 12500" "--workload c2" "--workload headline" "--nodes 6000"; do
  echo "=== $WL"
  for rep in 1 2; do
    echo -n "default (bx6i from 50k rows): "; run A=1
    echo -n "bx6i from 20k rows: "; run G4C_BX6I_MIN_ROWS=20000
    echo -n "bx6i from 10k rows: "; run G4C_BX6I_MIN_ROWS=10000
    echo -n "bx6i from 2k rows: "; run G4C_BX6I_MIN_ROWS=2000
  done
done
