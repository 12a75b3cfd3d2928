"""Where does the component-wise state error of an fp32-STORED covariance come from?  CPU only (oracle = the fp64 checker):
run the oracle for t0 frames, then step ONE more frame from (a) the exact P, (b) P rounded to fp32 everywhere, (c) P rounded
to fp32 except the per-feature 6x6 diagonal blocks and the 13 camera rows / columns, (d) except the diagonal blocks only.
Reports the component-wise error |d| / max(|ref|, 1e-4) of the feature parameters after that frame against (a)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as ol
from openekfmonoslam_amd.synth import SyntheticSequence

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kw = {}
if len(sys.argv) > 3:
    kw = eval(sys.argv[3])
ol.build()
seq = SyntheticSequence(N, t0 + 1, **kw)
o = ol.Oracle(seq.cam, seq.par, N + 8)
o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
for t in range(t0):
    o.step(*seq.frames[t], ol.ALGORITHMIC)
x, fp, P = o.x13(), o.feature_pos(), o.P()
n = P.shape[0]
r32 = lambda a: a.astype(np.float32).astype(np.float64)


def keep_mask(cam, blocks):
    m = np.zeros((n, n), dtype=bool)
    if cam:
        m[:13, :] = True
        m[:, :13] = True
    if blocks:
        for f in range(N):
            p = 13 + 6 * f
            m[p:p + 6, p:p + 6] = True
    return m


variants = {"exact": P, "all32": r32(P)}
for name, (cam, blk) in {"cross32(cam+blocks fp64)": (True, True), "blocks fp64 only": (False, True), "cam fp64 only": (True, False)}.items():
    m = keep_mask(cam, blk)
    variants[name] = np.where(m, P, r32(P))
res = {}
for name, Pv in variants.items():
    ov = ol.Oracle(seq.cam, seq.par, N + 8)
    ov.set_state(x, fp, seq.feature_type, seq.feature_desc, Pv)
    info = ov.step(*seq.frames[t0], ol.ALGORITHMIC)
    res[name] = (ov.x13(), ov.feature_pos(), (info.n_matches, info.n_inliers, info.n_rescued))
xe, fe, ie = res["exact"]
for name, (xv, fv, iv) in res.items():
    if name == "exact":
        continue
    e = np.abs(fv - fe) / np.maximum(np.abs(fe), 1e-4)
    k = np.unravel_index(e.argmax(), e.shape)
    srt = np.sort(e.reshape(-1))[::-1]
    print(f"N={N} after {t0} exact frames, frame {t0} from {name:28s}: componentwise max {e.max():.2e} (feature {k[0]} comp {k[1]} "
          f"value {fe[k]:.3e} err {abs(fv[k] - fe[k]):.2e}); 5 worst {[f'{v:.1e}' for v in srt[:5]]}; camera w rel "
          f"{np.abs(xv[10:13] - xe[10:13]).max() / np.abs(xe[10:13]).max():.2e}; decisions {'same' if iv == ie else 'DIFFER'}")
