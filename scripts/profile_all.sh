#!/bin/bash
# Kernel-trace profiles of the bench workloads (GPU box, from the repo root): scripts/profile_all.sh <outdir>
# writes <outdir>/kernel_stats_<workload>.csv (rocprofv3's own per-kernel statistics) and <outdir>/<workload>.json (the bench line
# of the profiled run).  Every run sits under `timeout`: rocprofv3 has been seen not to exit after writing its output.
out=$(realpath -m "$1"); root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for w in "n1000_f32x 40 10" "n1000_f32 40 10" "n200_f64 60 10" "n2000_f32x 12 3" "n5000_f32x 3 1"; do
    set -- $w
    timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$1" -- python3 "$root/bench.py" --workload $1 --steps $2 --warmup $3 --no-cpu-baseline --no-all-matched --no-fast-line > "$out/$1.json" 2> "$out/$1.err"
    f=$(find "$out/$1" -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && head -45 "$f" > "$out/kernel_stats_$1.csv"
    rm -rf "$out/$1"
done
