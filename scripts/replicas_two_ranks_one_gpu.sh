#!/bin/bash
# two replica ranks sharing cuda:0 over gloo: checks the N>1 JSON contract of bench.py (value = whole-job aggregate)
cd $GRAFT_REPO_ROOT
PORT=29533
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT EKF_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/rep_$r.log 2>&1 &
done
wait
tail -1 gpurun_out/rep_0.log | cut -c1-420; echo; tail -2 gpurun_out/rep_1.log | cut -c1-200
