import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from openekfmonoslam_amd import engine
from openekfmonoslam_amd.synth import SyntheticSequence
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
W, H = (1920, 1080) if N >= 3000 else (1280, 720)
seq = SyntheticSequence(N, 1, width=W, height=H)
res = {}
for prec in (2, 0):
    e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=prec)
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    i = e.step(*seq.frames[0])
    print(prec, i.n_matches, i.n_inliers, i.n_rescued, i.status, flush=True)
    x, fp, P = e.get_state()
    res[prec] = (np.diag(P).copy(), P[:, 5000].copy(), P[20000, :].copy())
    del P
    e.close()
d2, c2, r2 = res[2]; d0, c0, r0 = res[0]
bad = np.nonzero(np.abs(d2 - d0) > 1e-4)[0]
print("diag entries off:", len(bad), "of", len(d0), "first", bad[:10], "last", bad[-10:] if len(bad) else None)
if len(bad):
    print("bad mod 128 histogram (8 bins of 16):", np.bincount((bad % 128) // 16, minlength=8))
    print("bad tile index histogram (first 30 tiles):", np.bincount(bad // 128)[:30])
    k = bad[0]; print("example", k, d2[k], d0[k])
bc = np.nonzero(np.abs(c2 - c0) > 1e-6 * (1 + np.abs(c0)))[0]
print("column 5000 entries off:", len(bc), bc[:10], bc[-10:] if len(bc) else None)
br = np.nonzero(np.abs(r2 - r0) > 1e-6 * (1 + np.abs(r0)))[0]
print("row 20000 entries off:", len(br), br[:10], br[-10:] if len(br) else None)
