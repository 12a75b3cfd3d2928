"""The accuracy floor of fp32 STORAGE of the covariance, measured on the GPU: the fp64 engine run stage by stage (predict,
predictCameraMeasurements, match, ransac, update, rescue, update = EKF.cpp:273-532) with P rounded to fp32 at the points where
the fp32-storage configurations round it (upload, covariance prediction, each update) and nothing else -- against (a) the plain
fp64 engine (= the oracle to 1e-12, tests/test_gpu_parity*.py) and beside (b) the EKF_PRECISION_F32_EXACT engine and (c) the fast
EKF_PRECISION_F32 engine on the same frames.  Component-wise = every feature parameter against max(|own value|, 1e-4).
    python scripts/storage_floor_gpu.py N frames ["dict(width=1280, height=720)"]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from openekfmonoslam_amd import engine
from openekfmonoslam_amd.synth import SyntheticSequence
from parity_metric import block_errs

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw = eval(sys.argv[3]) if len(sys.argv) > 3 else ({"width": 1280, "height": 720} if N == 2000 else {"width": 1920, "height": 1080} if N >= 3000 else {})
seq = SyntheticSequence(N, F, **kw)
mk = lambda prec: engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=prec)


def staged(e, kps, desc, rnd):
    e.predict()
    rnd()
    e.predict_measurements()
    matches = e.match(kps, desc)
    mask, _ = e.ransac(matches)
    if mask.any():
        e.update(matches[mask])
        rnd()
    outl = matches[~mask]
    nres = 0
    if len(outl):
        p2, _, _ = e.predict_measurements(feat_idx=outl["featureIndex"])
        if len(p2):
            outl = outl[np.isin(outl["featureIndex"], p2["featureIndex"])]
            rm = e.rescue(outl)
            nres = int(rm.sum())
            if nres:
                e.update(outl[rm])
                rnd()
    return len(matches), int(mask.sum()), nres


engs = {"fp64": mk(0), "floor": mk(0), "exact": mk(2), "fast": mk(1)}
for e in engs.values():
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
engs["floor"].round_covariance_to_f32()
for t in range(F):
    kps, desc = seq.frames[t]
    i64 = engs["fp64"].step(kps, desc)
    dfl = staged(engs["floor"], kps, desc, engs["floor"].round_covariance_to_f32)
    iex = engs["exact"].step(kps, desc)
    ifa = engs["fast"].step(kps, desc)
    xr, fr, _ = engs["fp64"].get_state(want_P=False)
    line = [f"N={N} frame {t} (matches {i64.n_matches}, inliers {i64.n_inliers}, rescued {i64.n_rescued}; staged floor run: {dfl}; "
            f"exact {iex.n_inliers}/{iex.n_rescued}, fast {ifa.n_inliers}/{ifa.n_rescued}):"]
    for name in ("floor", "exact", "fast"):
        x, fp, _ = engs[name].get_state(want_P=False)
        be = block_errs(x, fp, xr, fr)
        er = np.abs(fp - fr) / np.maximum(np.abs(fr), 1e-4)
        k = np.unravel_index(er.argmax(), er.shape)
        line.append(f"  {name:5s}: component-wise {be['features_componentwise']:.2e} (feature {k[0]} comp {k[1]} value {fr[k]:.3e} err {abs(fp[k] - fr[k]):.2e}), "
                    f"rho block {be['feat_rho']:.2e}, w {be['w']:.2e}")
    print("\n".join(line), flush=True)
for e in engs.values():
    e.close()
