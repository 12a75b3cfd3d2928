"""How much of digit plane 0 of B is zero on the bench's own frames (N = 1000, EKF_PRECISION_F32_EXACT): per frame, the fraction of
16-row x 32-column pieces of plane 0 that hold anything, for the low-innovation update (stage calls: the hook sees the last update)
and for the high-innovation one (after the whole step).  python scripts/plane0_stats.py [N=1000] [frames=12]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openekfmonoslam_amd import engine  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seq = SyntheticSequence(N, F)
kw = dict(max_keypoints=len(seq.frames[0][0]) + 64, precision=2)
a, b = engine.EkfEngine(seq.cam, seq.par, N, **kw), engine.EkfEngine(seq.cam, seq.par, N, **kw)
for e in (a, b):
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
for t in range(F):
    # engine b: the frame stage by stage up to the first update
    b.predict()
    b.predict_measurements()
    m = b.match(*seq.frames[t])
    mask = b.ransac(m)[0] if len(m) else np.zeros(0, bool)
    inl = m[np.asarray(mask, bool)]
    if len(inl):
        b.update(inl)
        li = b.plane0_pieces()
    else:
        li = (0, 1)
    # engine a: the whole step; then b is put where a is
    i = a.step(*seq.frames[t])
    hi = a.plane0_pieces() if i.n_rescued else (0, 1)
    x, fp, P = a.get_state()
    b.set_state(x, fp, seq.feature_type, seq.feature_desc, P)
    print(f"frame {t}: LI update M={len(inl)}: {li[0]}/{li[1]} = {100.0 * li[0] / li[1]:.1f} % of the pieces of plane 0 non-zero; "
          f"HI update M={i.n_rescued}: {hi[0]}/{hi[1]} = {100.0 * hi[0] / hi[1]:.1f} %")
