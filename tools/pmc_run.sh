#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <what> <h> <c> <k>   (GPU box; one rocprofv3 pass per counter group)
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- python3 tools/pmc_one.py "$@" > $out/g$i.log 2>&1
done
ls $out
