mkdir -p gpurun_out/r2z
(
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_sharded.py tests/test_gpu_map_management.py tests/test_gpu_compat.py tests/test_gpu_sequence.py -q -x 2>&1 | grep -E "passed|failed|Error" | head -5
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 10 --no-all-matched --no-single-gpu-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1000', round(d['value'],1), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
python bench.py --workload n200_f64 --steps 40 --warmup 10 --no-all-matched --no-single-gpu-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n200', round(d['value'],1), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
python bench.py --workload n2000_f32 --steps 10 --warmup 5 --no-all-matched --no-single-gpu-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n2000', round(d['value'],1), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
) > gpurun_out/r2z/q.log 2>&1
cat gpurun_out/r2z/q.log
