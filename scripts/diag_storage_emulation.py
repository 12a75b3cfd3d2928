"""What does fp32 STORAGE of the covariance alone cost, when P is rounded to fp32 at exactly the points where the engine's
EKF_PRECISION_F32_EXACT configuration rounds it -- on upload, after the covariance prediction, after the low-innovation update,
after the high-innovation update -- and every operation in between is fp64?  CPU only: the fp64 oracle run stage by stage
(predict, predictCameraMeasurements, match, ransac, update, rescue, update: EKF.cpp:273-532 as restated by orc_step_impl), once
without and once with the roundings.  The difference is the floor of ANY fp32-storage engine on these frames; what the exact
engine shows above it is arithmetic.  (The earlier one-rounding-per-frame estimate, scripts/diag_p_rounding.py, misses the
rounding of the PREDICTED covariance before an update that shrinks variances by orders of magnitude: the absolute error of that
rounding survives the cancellation.)
    python scripts/diag_storage_emulation.py N frames "dict(width=1280, height=720)"
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as ol
from openekfmonoslam_amd.synth import SyntheticSequence

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kw = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
ol.build()
seq = SyntheticSequence(N, F, **kw)
r32 = lambda a: a.astype(np.float32).astype(np.float64)


def staged_frames(rounding):
    o = ol.Oracle(seq.cam, seq.par, N + 8)
    P0 = r32(seq.P0) if rounding else seq.P0
    o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)

    def rnd():
        if rounding:
            o.set_state(o.x13(), o.feature_pos(), seq.feature_type, seq.feature_desc, r32(o.P()))

    out = []
    for t in range(F):
        kps, desc = seq.frames[t]
        o.predict()
        rnd()
        preds, Hs, Hf = o.predict_measurements()
        matches = o.match(preds, kps, desc)
        pa, Hsa, Hfa = ol.align_to_matches(preds, Hs, Hf, matches)
        mask, _ = o.ransac(pa, Hsa, Hfa, matches)
        o.update(matches[mask], pa[mask], Hsa[mask], Hfa[mask], ol.ALGORITHMIC)
        rnd()
        outl = matches[~mask]
        nres = 0
        if len(outl):
            p2, Hs2, Hf2 = o.predict_measurements(feat_idx=outl["featureIndex"])
            if 0 < len(p2):
                keep = np.isin(outl["featureIndex"], p2["featureIndex"])
                outl = outl[keep]
                rm = o.rescue(outl, p2)
                nres = int(rm.sum())
                if nres:
                    o.update(outl[rm], p2[rm], Hs2[rm], Hf2[rm], ol.ALGORITHMIC)
                    rnd()
        out.append((o.x13(), o.feature_pos(), (len(matches), int(mask.sum()), nres)))
        print(f"  {'rounded' if rounding else 'exact  '} frame {t}: matches {len(matches)} inliers {int(mask.sum())} rescued {nres}", flush=True)
    return out


ref = staged_frames(False)
rnd = staged_frames(True)
for t in range(F):
    (xe, fe, ie), (xv, fv, iv) = ref[t], rnd[t]
    e = np.abs(fv - fe) / np.maximum(np.abs(fe), 1e-4)
    k = np.unravel_index(e.argmax(), e.shape)
    srt = np.sort(e.reshape(-1))[::-1]
    blk = {c: np.abs(fv[:, c] - fe[:, c]).max() / max(np.abs(fe[:, c]).max(), 1e-300) for c in (3, 4, 5)}
    print(f"N={N} frame {t}: fp32 STORAGE of P alone (4 roundings per frame, all arithmetic fp64): componentwise max {e.max():.2e} "
          f"(feature {k[0]} comp {k[1]} value {fe[k]:.3e} err {abs(fv[k] - fe[k]):.2e}); 5 worst {[f'{v:.1e}' for v in srt[:5]]}; "
          f"blocks theta {blk[3]:.1e} phi {blk[4]:.1e} rho {blk[5]:.1e}; camera w rel "
          f"{np.abs(xv[10:13] - xe[10:13]).max() / np.abs(xe[10:13]).max():.2e}; decisions {'same' if iv == ie else 'DIFFER'}")
