#!/bin/bash
# Runs the GPU test suite and a decision check of the bench while other filters hammer the same GPU: timing-dependent
# defects (missing in-launch or cross-launch dependencies) show up as differences under contention.
cd ${GRAFT_REPO_ROOT:-.}
(for i in 1 2 3; do python bench.py --workload n2000_f32 --no-cpu-baseline --no-roofline-pass --steps 40 --warmup 2 >/dev/null 2>&1; done) &
(for i in 1 2 3 4 5 6 7 8; do python bench.py --no-cpu-baseline --no-roofline-pass --steps 120 >/dev/null 2>&1; done) &
sleep 3
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
python bench.py --no-cpu-baseline --no-roofline-pass --steps 60 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('contended:', round(d['value'],1), c['mean_matches'], c['mean_li_inliers'], c['mean_rescued'], c['mean_ransac_hypotheses'])"
wait
python bench.py --no-cpu-baseline --no-roofline-pass --steps 60 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('alone:    ', round(d['value'],1), c['mean_matches'], c['mean_li_inliers'], c['mean_rescued'], c['mean_ransac_hypotheses'])"
