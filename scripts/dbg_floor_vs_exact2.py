"""After one frame: error of the covariance of (a) the storage floor (fp64 engine, P rounded to fp32 at the fp32 configurations'
rounding points) and (b) the EKF_PRECISION_F32_EXACT engine, each against the plain fp64 engine, in units of the entry's fp32 spacing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from openekfmonoslam_amd import engine
from openekfmonoslam_amd.synth import SyntheticSequence
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kw = {"width": 1280, "height": 720} if N >= 1400 else {}
seq = SyntheticSequence(N, 1, **kw)
mk = lambda prec: engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=prec)
fl, ex, rf = mk(0), mk(2), mk(0)
for e in (fl, ex, rf):
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
fl.round_covariance_to_f32()
kps, desc = seq.frames[0]
rnd = fl.round_covariance_to_f32
fl.predict(); rnd(); fl.predict_measurements(); m = fl.match(kps, desc); mask, _ = fl.ransac(m)
fl.update(m[mask]); rnd()
out = m[~mask]; p2, _, _ = fl.predict_measurements(feat_idx=out["featureIndex"]); out = out[np.isin(out["featureIndex"], p2["featureIndex"])]
rm = fl.rescue(out); fl.update(out[rm]); rnd()
ex.step(kps, desc)
rf.step(kps, desc)
_, _, Pr = rf.get_state()
ulp = np.spacing(np.abs(Pr).astype(np.float32)).astype(np.float64)
for name, e in (("floor", fl), ("exact", ex)):
    _, _, P = e.get_state()
    r = np.abs(P - Pr) / ulp
    del P
    for lo, hi in ((0, 1e-12), (1e-12, 1e-9), (1e-9, 1e-6), (1e-6, 1e-3), (1e-3, 10)):
        sel = (np.abs(Pr) >= lo) & (np.abs(Pr) < hi)
        if sel.any():
            rs = r[sel]
            print(f"{name} |P| in [{lo:g}, {hi:g}) ({sel.mean():.3f} of entries): error in spacings: rms {np.sqrt((rs ** 2).mean()):.2f}  99.9th pct {np.percentile(rs, 99.9):.1f}  max {rs.max():.1f}", flush=True)
