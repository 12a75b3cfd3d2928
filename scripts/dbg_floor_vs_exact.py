"""After one frame: how does the covariance of the EKF_PRECISION_F32_EXACT engine differ from the storage floor's (fp64 engine with
P rounded to fp32 at the same points)?  Both hold fp32 values; differences are counted in units of the fp32 spacing of the entry."""
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
fl, ex = mk(0), mk(2)
for e in (fl, ex):
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
fl.round_covariance_to_f32()
kps, desc = seq.frames[0]
rnd = fl.round_covariance_to_f32
fl.predict(); rnd(); fl.predict_measurements(); m = fl.match(kps, desc); mask, _ = fl.ransac(m)
fl.update(m[mask]); rnd()
out = m[~mask]; p2, _, _ = fl.predict_measurements(feat_idx=out["featureIndex"]); out = out[np.isin(out["featureIndex"], p2["featureIndex"])]
rm = fl.rescue(out); fl.update(out[rm]); rnd()
ex.step(kps, desc)
_, _, Pf = fl.get_state()
_, _, Pe = ex.get_state()
d = np.abs(Pe - Pf)
ulp = np.spacing(np.abs(Pf).astype(np.float32)).astype(np.float64)
r = d / ulp
print(f"N={N}: entries equal {np.mean(d == 0):.4f}; differing by <= 1 spacing {np.mean((r > 0) & (r <= 1)):.4f}; by more {np.mean(r > 1):.6f}; max {r.max():.1f} spacings; max abs {d.max():.2e}")
i, j = np.unravel_index(r.argmax(), r.shape)
print("worst entry", i, j, Pf[i, j], Pe[i, j])
for lo, hi in ((0, 1e-9), (1e-9, 1e-6), (1e-6, 1e-3), (1e-3, 10)):
    sel = (np.abs(Pf) >= lo) & (np.abs(Pf) < hi)
    if sel.any():
        print(f"  |P| in [{lo:g}, {hi:g}): {sel.mean():.3f} of entries, unequal {np.mean(d[sel] > 0):.4f}, > 1 spacing {np.mean(r[sel] > 1):.6f}, max {r[sel].max():.1f}")
# where are the largest absolute differences?
flat = np.argsort(d, axis=None)[::-1][:12]
for f in flat:
    i, j = np.unravel_index(f, d.shape)
    fi, fj = (i - 13) // 6 if i >= 13 else -1, (j - 13) // 6 if j >= 13 else -1
    print(f"  |diff| {d[i, j]:.2e} at ({i},{j}) = feature {fi} comp {(i - 13) % 6 if i >= 13 else i} x feature {fj} comp {(j - 13) % 6 if j >= 13 else j}: floor {Pf[i, j]:.6e} exact {Pe[i, j]:.6e}")
big = d > 100 * ulp
rows = np.bincount(np.nonzero(big)[0], minlength=d.shape[0])
print("rows with most entries off by > 100 spacings:", np.argsort(rows)[::-1][:10], rows[np.argsort(rows)[::-1][:10]])
print("entries off by > 100 spacings by class: camera rows/cols", int(big[:13, :].sum() + big[13:, :13].sum()), " same-feature blocks",
      int(sum(big[13 + 6 * f:19 + 6 * f, 13 + 6 * f:19 + 6 * f].sum() for f in range(N))), " total", int(big.sum()))
matched = set(int(x) for x in m["featureIndex"])
top_rows = np.argsort(rows)[::-1][:10]
print("are the features of those rows matched in this frame?", [((r - 13) // 6, ((r - 13) // 6) in matched) for r in top_rows if r >= 13])
