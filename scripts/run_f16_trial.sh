mkdir -p gpurun_out/f16
V=$PWD/graphs4cfd_amd/lib_variants
( echo "--- default"; timeout 300 python scripts/step_breakdown.py 2>&1 | tail -50
  echo "--- weights always from the same 2 KB (L1 hits; wrong results, timing only)"; G4C_LIB_PATH=$V/libg4c_l1w.so timeout 300 python scripts/step_breakdown.py 2>&1 | tail -50
) > gpurun_out/f16/run13.log 2>&1
