#!/bin/bash
# Same-box A/B of two environments on bench.py (box-to-box / DVFS noise between separate gpurun calls is ~3 %):
#   bash scripts/ab_bench.sh "G4C_LIB_PATH=$PWD/graphs4cfd_amd/lib/libg4c_old.so" "" [bench.py arguments...]
# runs three interleaved rounds of `env <A> python bench.py ...` and `env <B> python bench.py ...` and prints the values.
A=$1; B=$2; shift 2
for i in 1 2 3; do for v in A B; do
  if [ $v = A ]; then E=$A; else E=$B; fi
  env $E python bench.py --no-cpu-baseline --no-strict-range --no-roofline --no-partition-check "$@" 2>/dev/null | tail -1 \
    | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v [$E]',round(d['value'],2))"
done; done
