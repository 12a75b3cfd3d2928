"""Per-block error of the fp32-covariance engine against the fp64 oracle, frame by frame, for several scenes
(GPU box).  usage: parity_blocks.py [N] [frames] [scene ...]   scene = horizon:<h> | seed:<s> | default"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_lib as ol  # noqa: E402
from openekfmonoslam_amd import engine  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scenes = sys.argv[3:] or ["horizon:25", "horizon:70", "default"]
for sc in scenes:
    kw = {}
    if sc.startswith("horizon:"):
        kw["horizon"] = int(sc.split(":")[1])
    elif sc.startswith("seed:"):
        kw["seed"] = int(sc.split(":")[1], 0)
    seq = SyntheticSequence(N, F, **kw)
    e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=2 * N + 64, precision=1)
    o = ol.Oracle(seq.cam, seq.par, N + 8)
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    for t in range(F):
        gi = e.step(*seq.frames[t])
        oi = o.step(*seq.frames[t], ol.ALGORITHMIC)
        x, fp, P = e.get_state()
        xo, fpo, Po = o.x13(), o.feature_pos(), o.P()
        blk = {k: float(np.abs(x[s] - xo[s]).max() / max(np.abs(xo[s]).max(), 1e-9)) for k, s in
               (("r", slice(0, 3)), ("q", slice(3, 7)), ("v", slice(7, 10)), ("w", slice(10, 13)))}
        fe = np.abs(fp - fpo) / np.maximum(np.abs(fpo), 1e-4)
        comp = fe.max(axis=0)
        fi = np.unravel_index(fe.argmax(), fe.shape)
        same = (gi.n_matches, gi.n_inliers, gi.n_rescued, gi.n_hypotheses) == (oi.n_matches, oi.n_inliers, oi.n_rescued, oi.n_hypotheses)
        print(f"{sc} frame {t}: same={same} li={oi.n_inliers} hi={oi.n_rescued} | " +
              " ".join(f"{k}={v:.1e}" for k, v in blk.items()) +
              f" | feat max {fe.max():.2e} (feature {fi[0]} comp {fi[1]} value {fpo[fi]:.3e} abs err {abs(fp[fi] - fpo[fi]):.1e})"
              f" per-comp {np.array2string(comp, precision=1)}"
              f" | P max {np.abs(P - Po).max() / np.abs(Po).max():.1e} fro {np.linalg.norm(P - Po) / np.linalg.norm(Po):.1e}", flush=True)
    e.close()
