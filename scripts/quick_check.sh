cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_exact_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -5)
for w in $WORKLOADS; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-all-matched --no-fast-line --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w updates/s %.1f  ms/frame %.4f  sweep us/panel %.2f  parity %s' % (d['value'], d['ms_per_step'], d['roofline_sweep']['us_per_panel'], d.get('parity_ok')), d.get('stage_ms_per_step'))"
done
if [ -n "$TRACE" ]; then EKF_ENGINE_LIB=variants/libekf_engine_trace.so timeout 200 python scripts/persist_trace.py 1000 12 0 2>&1 | grep -v amdgpu.ids | head -n 34 | cut -c1-150; fi
