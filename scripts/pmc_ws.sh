#!/bin/bash
# PMC passes over the level-1 message launch alone (scripts/ws_pmc.py): pipe utilisation, LDS, wait buckets, L2.
# Usage (GPU box): bash scripts/pmc_ws.sh <ws|bx6i|tile> <tag> [passes...]     -> gpurun_out/pmc_<tag>_<pass>/ + a summary on stdout
K=$1; TAG=$2; shift 2
PASSES=${@:-"util sq3 lds sq2 tcc"}
EXTRA=${PMC_EXTRA_ARGS:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# which kernel sources the counters below were collected from (bench.py compares this with the sources it runs: roofline.mfma_util_pmc_imported)
KERN=mlp_ws_kernel; [ "$K" = "tile" ] && KERN=mlp_bx6_kernel; [ "$K" = "bx6i" ] && KERN=mlp_bx6i_kernel
echo "# kernel: $KERN  source_sha16: $(python -c "import bench; print(bench.source_sha16('$KERN'))")"
declare -A C
C[util]="MfmaUtil VALUBusy MemUnitStalled"
C[sq3]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES"
C[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
C[sq2]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"
C[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
# dynamic instruction mix of the vector ALU, and how many cycles the vector and matrix pipes run at the same time
C[mix]="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
C[mix2]="SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM"
C[coexec]="SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
for p in $PASSES; do
  OUT=gpurun_out/pmc_${TAG}_$p; rm -rf $OUT
  timeout 300 rocprofv3 --pmc ${C[$p]} --kernel-trace --output-format csv -d $OUT -o p -- python scripts/ws_pmc.py $K $EXTRA > $OUT.log 2>&1
  echo "== $TAG $p"; python scripts/pmc_summary.py $OUT | grep -A14 "mlp_"
done
