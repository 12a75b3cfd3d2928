#!/bin/bash
# Everything profiles/ holds for a round, from the repo root on the GPU box: scripts/round_artifacts.sh <outdir> [skip-tests]
# (every profiler run sits under `timeout`; copy <outdir>/* to profiles/<round>_* afterwards: scripts/collect_profiles.sh)
out=$(realpath -m "$1"); root=$(pwd); mkdir -p "$out"
if [ -z "$2" ]; then
  timeout 2900 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1
  tail -3 "$out/pytest_gpu.log"
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1
B="timeout 600 python bench.py"
$B > "$out/bench_n1000_f32x.json" 2> "$out/bench_n1000_f32x.err"
$B --workload n1000_f32 > "$out/bench_n1000_f32.json" 2> /dev/null
$B --workload n200_f64 --steps 60 --warmup 10 > "$out/bench_n200_f64.json" 2> /dev/null
$B --workload n2000_f32x --steps 20 --warmup 5 > "$out/bench_n2000_f32x.json" 2> /dev/null
$B --workload n2000_f32 --steps 20 --warmup 5 > "$out/bench_n2000_f32.json" 2> /dev/null
$B --workload n2000_auto --steps 20 --warmup 5 > "$out/bench_n2000_auto.json" 2> /dev/null
$B --workload n5000_f32x --steps 3 --warmup 1 > "$out/bench_n5000_f32x.json" 2> /dev/null
$B --workload n5000_f64x --steps 3 --warmup 1 > "$out/bench_n5000_f64x.json" 2> /dev/null
$B --workload n5000_f32 --steps 3 --warmup 1 > "$out/bench_n5000_f32.json" 2> /dev/null
$B --workload n1000_f32x --sweep-mode 4 > "$out/bench_n1000_f32x_launches.json" 2> /dev/null
$B --workload n200_f64 --sweep-mode 4 --steps 60 --warmup 10 > "$out/bench_n200_f64_launches.json" 2> /dev/null
$B --matcher ncc --workload n2000_f32x --steps 20 --warmup 5 > "$out/bench_n2000_f32x_ncc.json" 2> /dev/null
$B --emulate-shards 4 --workload n2000_f32x --steps 6 --warmup 2 --no-cpu-baseline > "$out/bench_n2000_f32x_emulated4.json" 2> /dev/null
# round 6: how much of digit plane 0 is zero on the bench's frames; two processes sharing the GPU (no error may surface)
timeout 300 python scripts/plane0_stats.py 1000 30 2>/dev/null | grep -v amdgpu.ids > "$out/plane0_pieces.txt"
timeout 600 python scripts/two_processes_one_gpu.py 200 1000 2 > "$out/two_processes_one_gpu.txt" 2>&1
# the downdate kernel alone: variants (0 persistent, 1 first version), bitwise check against the CPU, ablations at m = 298 / 1014 when built
bash scripts/pu_i8_micro.sh > "$out/pu_i8_bench.txt" 2>&1
# timelines of the persistent sweep: only when the debug build is there (scripts/build_trace_variant.sh)
if [ -f variants/libekf_engine_trace.so ]; then
  EKF_ENGINE_LIB=variants/libekf_engine_trace.so timeout 300 python scripts/persist_trace.py 1000 12 0 2>/dev/null | grep -v amdgpu.ids > "$out/persist_trace_n1000_f32x.txt"
  EKF_ENGINE_LIB=variants/libekf_engine_trace.so timeout 300 python scripts/persist_trace.py 1000 12 1 2>/dev/null | grep -v amdgpu.ids > "$out/persist_trace_n1000_f32x_hi.txt"
  EKF_ENGINE_LIB=variants/libekf_engine_trace.so PRECISION=0 timeout 300 python scripts/persist_trace.py 200 12 0 2>/dev/null | grep -v amdgpu.ids > "$out/persist_trace_n200_f64.txt"
fi
bash scripts/profile_all.sh "$out/prof"
mv "$out"/prof/kernel_stats_*.csv "$out"/ 2>/dev/null
bash scripts/pmc_p_update.sh "$out/pmc"
dbs=""
for i in 1 2 3 4 7; do d=$(find $out/pmc/pass$i -name '*.db' 2>/dev/null | head -1); [ -n "$d" ] && dbs="$dbs $d"; done
export PMC_KERNEL_LIKE="%k_p_update_i8p%"
python3 scripts/profile_summary.py pmc_sq "$out/pmc_sq_p_update_n1000_f32x.csv" $dbs
python3 scripts/profile_summary.py pmc "$(find $out/pmc/pass5 -name '*.db' | head -1)" "$(find $out/pmc/pass6 -name '*.db' | head -1)" "$out/pmc_hbm_p_update_n1000_f32x.csv" 20
rm -rf "$out"/pmc/pass*/
tail -c 600 "$out/bench_n1000_f32x.json"
