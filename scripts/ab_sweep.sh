mkdir -p gpurun_out/r2z
(
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
for v in 100000 200 100 400 100000 200 100 400; do
  EKF_TILES_FIRST=$v python bench.py --steps 40 --warmup 10 --no-all-matched --no-single-gpu-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles_first>=$v', round(d['value'],1), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
) > gpurun_out/r2z/tf.log 2>&1
cat gpurun_out/r2z/tf.log
