"""P-update kernel time as a function of the number of matches in one update (GPU box): fixed N, matches taken from
the predictions, one ekf_update per size; prints m, kernel ms, TFLOP/s.  Shows the fixed cost of a launch (m -> 0)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from openekfmonoslam_amd import engine
from openekfmonoslam_amd.ekftypes import MATCH_DTYPE
from openekfmonoslam_amd.synth import SyntheticSequence
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seq = SyntheticSequence(N, 1)
e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=2 * N + 64, precision=1)
rng = np.random.default_rng(0)
n = 13 + 6 * N
for M in (8, 16, 32, 64, 96, 128, 160, 256, 384, 512, 768):
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    e.predict()
    preds, _, _ = e.predict_measurements()
    m = np.zeros(M, dtype=MATCH_DTYPE)
    m["featureIndex"] = preds["featureIndex"][:M]
    m["imagePos"] = preds["imagePos"][:M] + rng.normal(0, 0.3, (M, 2))
    e.update(m)       # first update after set_state runs the AVG variant: warm up with it
    e.predict()
    preds, _, _ = e.predict_measurements()
    m["imagePos"] = preds["imagePos"][:M] + rng.normal(0, 0.3, (M, 2))
    e.timing(True)
    e.timing_reset()
    for _ in range(5):
        e.update(m)
    mm, ms = e.p_update_launches()
    t = float(np.median(ms))
    print(f"m {2 * M:5d}  kernel {t * 1e3:7.1f} us  {n * n * float(mm[0]) / (t * 1e-3) / 1e12:6.1f} TFLOP/s  P traffic {2 * n * n * 4 / (t * 1e-3) / 1e12:5.2f} TB/s")
