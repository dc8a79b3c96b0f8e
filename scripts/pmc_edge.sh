#!/bin/bash
# PMC passes over the hoisted edge MLP alone (scripts/ab_test.py with one library): SQ / TA / TCP / TCC groups
# Usage (GPU box): bash scripts/pmc_edge.sh <lib.so> <tag> [passes...]
LIB=$1; TAG=$2; shift 2
PASSES=${@:-"sq ta tcp tcc"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A C
C[sq]="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM"
C[ta]="TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
C[tcp]="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
C[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum"
C[util]="MfmaUtil VALUBusy MemUnitStalled"
C[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
C[sq2]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"
C[sq3]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES"
for p in $PASSES; do
  OUT=gpurun_out/pmc_edge_${TAG}_$p; rm -rf $OUT
  timeout 300 rocprofv3 --pmc ${C[$p]} --kernel-trace --output-format csv -d $OUT -o p -- python scripts/ab_test.py $LIB --rows 600000 --rounds 3 --inner 2 --precision ${G4C_PMC_PRECISION:-bf16x6} > $OUT.log 2>&1
  echo "== $TAG $p"; python scripts/pmc_summary.py $OUT | grep -A12 "mlp_"
done
