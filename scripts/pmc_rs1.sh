cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A C
C[util]="MfmaUtil VALUBusy MemUnitStalled"
C[util2]="MemUnitBusy WriteUnitStalled LDSBankConflict"
C[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
C[sq2]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE"
C[sq3]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES"
C[ta]="TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
C[tcp]="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
C[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum"
C[tcc2]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
for p in util util2 lds sq2 sq3 ta tcp tcc tcc2; do
  OUT=gpurun_out/pmc_rs1_$p; rm -rf $OUT
  timeout 200 rocprofv3 --pmc ${C[$p]} --kernel-trace --output-format csv -d $OUT -o p -- python scripts/rs1_time.py > $OUT.log 2>&1
  echo "== rs1 $p"; python scripts/pmc_summary.py $OUT | grep -A12 "mlp_rs1"
done
