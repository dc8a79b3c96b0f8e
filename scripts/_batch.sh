cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/b1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "range or static_encoder or rollouts or warns" 2>&1 | tail -15 > gpurun_out/b1/pytest_subset.log
bash scripts/ws_ab_variants.sh shipped poly prio > gpurun_out/b1/ws_ab.log 2>&1
for v in poly prio; do
  G4C_LIB_PATH=$PWD/graphs4cfd_amd/lib/libg4c_ws_$v.so bash scripts/pmc_ws.sh ws r05_$v util sq3 coexec > gpurun_out/b1/pmc_$v.txt 2>&1
done
bash scripts/pmc_ws.sh ws r05_shipped util sq3 coexec > gpurun_out/b1/pmc_shipped.txt 2>&1
