#!/bin/bash
# Everything profiles/ holds for a round, from the repo root on the GPU box: scripts/round_artifacts.sh <outdir>
out=$(realpath -m "$1"); root=$(pwd); mkdir -p "$out"
python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1
python bench.py > "$out/bench_n1000_f32.json" 2> "$out/bench_n1000_f32.err"
python bench.py --workload n200_f64 --steps 60 --warmup 10 > "$out/bench_n200_f64.json" 2> /dev/null
python bench.py --workload n1000_f64 --steps 20 --warmup 5 > "$out/bench_n1000_f64.json" 2> /dev/null
python bench.py --workload n2000_f32 --steps 20 --warmup 5 > "$out/bench_n2000_f32.json" 2> /dev/null
python bench.py --workload n5000_f32 --steps 3 --warmup 1 > "$out/bench_n5000_f32.json" 2> /dev/null
python bench.py --workload n5000_f64 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_n5000_f64.json" 2> /dev/null
python bench.py --matcher ncc --workload n2000_f32 --steps 20 --warmup 5 > "$out/bench_n2000_f32_ncc.json" 2> /dev/null
scripts/micro/pu_bench 1000 298,1014,2000 15 > "$out/pu_bench.txt" 2>&1
# per-role sweep traces: only when the debug build is there (scripts/build_trace_variant.sh)
if [ -f variants/libekf_engine_trace.so ]; then
  EKF_ENGINE_LIB=variants/libekf_engine_trace.so SWEEP_MODE=1 python scripts/sweep_trace.py 1000 15 2>/dev/null | grep -v amdgpu.ids > "$out/sweep_trace_n1000_f32.txt"
  EKF_ENGINE_LIB=variants/libekf_engine_trace.so SWEEP_MODE=0 python scripts/sweep_trace.py 1000 15 2>/dev/null | grep -v amdgpu.ids > "$out/sweep_trace_pairs_n1000_f32.txt"
  EKF_ENGINE_LIB=variants/libekf_engine_trace.so SWEEP_MODE=2 python scripts/sweep_trace.py 2000 6 2>/dev/null | grep -v amdgpu.ids > "$out/sweep_trace_n2000_f32.txt"
fi
EKF_ENGINE_LIB=variants/libekf_engine_trace.so python scripts/contention_probe.py 30 2>/dev/null | grep -v amdgpu.ids > "$out/contention_probe.txt"
python scripts/diag_n5000_paths.py 0 1 2>/dev/null | grep -E "^path|^   " > "$out/n5000_three_frames_fp32.txt"
bash scripts/profile_all.sh "$out/prof"
for w in n1000_f32 n200_f64 n2000_f32 n5000_f32; do
  db=$(find "$out/prof/$w" -name "*.db" | head -1)
  [ -n "$db" ] && python3 scripts/profile_summary.py stats "$db" "$out/kernel_stats_$w.csv"
  rm -rf "$out/prof/$w"
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$out/prof/ncc" -- python3 "$root/bench.py" --matcher ncc --workload n2000_f32 --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 )
db=$(find "$out/prof/ncc" -name "*.db" | head -1)
[ -n "$db" ] && python3 scripts/profile_summary.py stats "$db" "$out/kernel_stats_n2000_f32_ncc.csv"
rm -rf "$out/prof/ncc"
bash scripts/pmc_p_update.sh "$out/pmc"
dbs=""
for i in 1 2 3 4; do dbs="$dbs $(find $out/pmc/pass$i -name '*.db' | head -1)"; done
python3 scripts/profile_summary.py pmc_sq "$out/pmc_sq_p_update_n1000_f32.csv" $dbs
python3 scripts/profile_summary.py pmc "$(find $out/pmc/pass5 -name '*.db' | head -1)" "$(find $out/pmc/pass6 -name '*.db' | head -1)" "$out/pmc_hbm_p_update_n1000_f32.csv" 21
rm -rf "$out/pmc"/pass*/ 
tail -3 "$out/pytest_gpu.log"; tail -c 600 "$out/bench_n1000_f32.json"
