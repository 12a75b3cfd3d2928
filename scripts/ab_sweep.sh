mkdir -p gpurun_out/r2z
(
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
  python bench.py --steps 40 --warmup 10 --no-all-matched --no-single-gpu-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1000', round(d['value'],1), d.get('parity_ok'), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()}, d['roofline']['frac'])"
done
python bench.py --workload n200_f64 --steps 40 --warmup 10 --no-all-matched --no-single-gpu-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n200', round(d['value'],1), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
python bench.py --workload n2000_f32 --steps 10 --warmup 5 --no-all-matched --no-single-gpu-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n2000', round(d['value'],1), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
python bench.py --workload n5000_f32 --steps 3 --warmup 1 --no-all-matched --no-single-gpu-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n5000', round(d['value'],2), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
) > gpurun_out/r2z/full.log 2>&1
tail -30 gpurun_out/r2z/full.log
