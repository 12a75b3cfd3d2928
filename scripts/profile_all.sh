#!/bin/bash
# Kernel-trace profiles of the bench workloads (GPU box, from the repo root): scripts/profile_all.sh <outdir>
out=$(realpath "$1"); root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for w in "n1000_f32 40 10" "n200_f64 60 10" "n2000_f32 12 3" "n5000_f32 3 1"; do
    set -- $w
    rocprofv3 --kernel-trace --stats -d "$out/$1" -- python3 "$root/bench.py" --workload $1 --steps $2 --warmup $3 --no-cpu-baseline --no-all-matched > "$out/$1.json" 2> "$out/$1.err"
done
