#!/bin/bash
# rocprofv3 PMC passes for k_p_update (separate passes, counters only -- no tracing), default bench workload.
# usage (GPU box, from the repo root): scripts/pmc_p_update.sh <outdir>
# the bench's own frames: 10 warm-up + 20 timed (profile_summary.py drops the warm-up launches)
out=$(realpath "$1"); root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-include-regex "k_p_update" -d "$out/pass$i" -- python3 "$root/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --no-roofline-pass --no-all-matched > /dev/null 2> "$out/pass$i.err"
done
