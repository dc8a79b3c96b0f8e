cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/b16
bash scripts/ws_ab_variants.sh shipped ah3 ah4 > gpurun_out/b16/ws_ab.log 2>&1
for r in 1 2; do for v in shipped ah3 ah4; do
  if [ "$v" = shipped ]; then L=graphs4cfd_amd/lib/libg4c.so; else L=graphs4cfd_amd/lib/libg4c_ws_$v.so; fi
  G4C_LIB_PATH=$PWD/$L timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v c3',round(d['value'],2))"
done; done > gpurun_out/b16/c3_ab.log 2>&1
