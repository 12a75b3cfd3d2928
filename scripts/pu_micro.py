"""Micro-benchmark of the P-update kernel alone (GPU box): repeated ekf_update on a fixed state, reports kernel ms."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from openekfmonoslam_amd import engine
from openekfmonoslam_amd.synth import SyntheticSequence
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seq = SyntheticSequence(N, 14)
e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=2*N+64, precision=1)
e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
e.upload_frames(seq.frames)
e.timing(True)
for t in range(14):
    try:
        e.step_frame(t)
    except Exception as ex:
        print("step", t, ex); break
m, ms = e.p_update_launches()
n = 13 + 6 * N
for a, b in zip(m, ms):
    print(int(a), round(float(b), 4), round(n * n * float(a) / (b * 1e-3) / 1e12, 1))
