cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_exact_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -5)
for w in n1000_f32x n200_f64 n2000_f32x; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-all-matched --no-fast-line --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w updates/s %.1f  ms/frame %.4f  sweep us/panel %.2f  parity %s' % (d['value'], d['ms_per_step'], d['roofline_sweep']['us_per_panel'], d.get('parity_ok')), d.get('stage_ms_per_step'))"
done
