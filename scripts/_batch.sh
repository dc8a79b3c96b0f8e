cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/b14
G4C_LIB_PATH=$PWD/graphs4cfd_amd/lib/libg4c_ws_skew.so timeout 900 python scripts/ws_check.py --stress 20 2>&1 | grep -v "^ok\|amdgpu.ids" | tail -6 > gpurun_out/b14/ws_check_skew.log
bash scripts/ws_ab_variants.sh shipped skew > gpurun_out/b14/ws_ab.log 2>&1
for r in 1 2; do for v in shipped skew; do
  if [ "$v" = shipped ]; then L=graphs4cfd_amd/lib/libg4c.so; else L=graphs4cfd_amd/lib/libg4c_ws_$v.so; fi
  G4C_LIB_PATH=$PWD/$L timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-strict-range 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v headline',round(d['value'],2),'c2',round(d['configs']['c2']['value'],1),'c3',round(d['configs']['c3']['value'],2))"
done; done > gpurun_out/b14/bench_ab.log 2>&1
