#!/bin/bash
# quick GPU check of the persistent sweep: correctness subset, bench lines of library variants, one trace
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_exact_edge_cases.py -x -q -k "across_block_boundaries and sweep" 2>&1 | tail -3) > gpurun_out/r05_ps_t1.txt
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3) > gpurun_out/r05_ps_t2.txt
tail -n 2 gpurun_out/r05_ps_t1.txt gpurun_out/r05_ps_t2.txt
for lib in "$@"; do
  echo "== $lib"
  EKF_ENGINE_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-all-matched --no-fast-line --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('updates/s %.1f  ms/frame %.4f  sweep us/panel %.2f  sweep ms/frame %.4f' % (d['value'], d['ms_per_step'], d['roofline_sweep']['us_per_panel'], d['roofline_sweep']['ms_per_frame']))"
done
EKF_ENGINE_LIB=variants/libekf_engine_trace.so timeout 200 python scripts/persist_trace.py 1000 12 0 > gpurun_out/r05_ps_trace_li.txt 2>&1
tail -n 31 gpurun_out/r05_ps_trace_li.txt
