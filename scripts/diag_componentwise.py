"""Component-wise state parity (|d| / max(|ref|, 1e-4)) of the fp32-covariance engine against the fp64 oracle, per update
path (0 = by size, 2 = inverse + GEMM), for the N = 1000 scenes of tests/test_gpu_parity_large.py.  GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as ol
from openekfmonoslam_amd import engine
from openekfmonoslam_amd.synth import SyntheticSequence
from parity_metric import parity_report

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
paths = [int(p) for p in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 2]
precs = [int(p) for p in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1]   # 1 fp32, 2 fp32 storage + exact downdate
paths = [(p, q) for q in precs for p in paths]
ol.build()
scenes = [("scene25", dict(horizon=25)), ("scene70", dict(horizon=70)), ("scene100", dict()), ("seedA", dict(seed=0xC0FFEE)),
          ("seedB", dict(seed=20260102))]
if N != 1000:
    scenes = [("default", dict(width=1280, height=720) if N == 2000 else dict())]
if os.environ.get("EKF_DIAG_KW"):  # e.g. "dict(width=1280, height=720)"
    scenes = [("custom", eval(os.environ["EKF_DIAG_KW"]))]
for name, kw in scenes:
    seq = SyntheticSequence(N, F, **kw)
    o = ol.Oracle(seq.cam, seq.par, N + 8)
    o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    engs = []
    for p, prec in paths:
        e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=prec)
        e.set_update_path(p)
        if os.environ.get("EKF_DIAG_SWEEP_MODE"):
            e.set_sweep_mode(int(os.environ["EKF_DIAG_SWEEP_MODE"]))
        e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
        engs.append(e)
    for t in range(F):
        o.step(*seq.frames[t], ol.ALGORITHMIC)
        xo, fpo, Po = o.x13(), o.feature_pos(), o.P()
        for (p, prec), e in zip(paths, engs):
            e.step(*seq.frames[t])
            x, fp, P = e.get_state()
            be = parity_report(x, fp, P, xo, fpo, Po)
            fe = np.abs(fp - fpo) / np.maximum(np.abs(fpo), 1e-4)
            fi = np.unravel_index(fe.argmax(), fe.shape)
            print(f"{name} frame {t} path {p} precision {prec}: componentwise {be['features_componentwise']:.2e} (feature {fi[0]} comp {fi[1]} "
                  f"value {fpo[fi]:.3e} err {abs(fp[fi] - fpo[fi]):.2e})  rho {be['feat_rho']:.2e} w {be['w']:.2e} P {be['P_max']:.2e}", flush=True)
    for e in engs:
        e.close()
