#!/bin/bash
# rocprofv3 PMC passes for the downdate kernel of the default bench workload (k_p_update_i8p, exact configuration): separate
# passes, counters only -- no tracing.  usage (GPU box, from the repo root): scripts/pmc_p_update.sh <outdir> [workload] [regex]
# the bench's own frames: 10 warm-up + 20 timed (profile_summary.py drops the warm-up launches)
out=$(realpath -m "$1"); root=$(pwd); wl=${2:-n1000_f32x}; rx=${3:-k_p_update_i8p}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_I8"; do
    i=$((i + 1))
    timeout 240 rocprofv3 --pmc $set --kernel-include-regex "$rx" -d "$out/pass$i" -- python3 "$root/bench.py" --workload $wl --steps 20 --warmup 10 --no-cpu-baseline --no-roofline-pass --no-all-matched --no-fast-line > /dev/null 2> "$out/pass$i.err"
done
