for i in 1 2 3; do for v in old new; do
  if [ $v = old ]; then export G4C_LIB_PATH=$PWD/graphs4cfd_amd/lib/libg4c_old.so; else unset G4C_LIB_PATH; fi
  python bench.py --no-cpu-baseline --no-strict-range --no-roofline --no-partition-check $@ 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v',round(d['value'],2))"
done; done
