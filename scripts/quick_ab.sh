cd $GRAFT_REPO_ROOT
# A/B of library variants on one box: LIBS="a.so b.so" [WORKLOADS="..."] bash scripts/quick_ab.sh
for rep in 1 2; do
for lib in $LIBS; do
  for w in ${WORKLOADS:-n1000_f32x n200_f64}; do
  EKF_ENGINE_LIB=$lib timeout 300 python bench.py --workload $w --steps 40 --warmup 8 --no-all-matched --no-fast-line --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $w updates/s %.1f  ms/frame %.4f  sweep us/panel %.2f' % (d['value'], d['ms_per_step'], d['roofline_sweep']['us_per_panel']), {k: round(v, 4) for k, v in d['stage_ms_per_step'].items()})"
  done
done
done
