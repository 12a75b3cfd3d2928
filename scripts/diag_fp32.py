"""Diagnostic (GPU box): error of the fp32-covariance engine vs the fp64 oracle, per frame."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from openekfmonoslam_amd import engine
from openekfmonoslam_amd.synth import SyntheticSequence

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 1
seq = SyntheticSequence(N, F)
e = engine.EkfEngine(seq.cam, seq.par, N + 8, max_keypoints=4 * N + 64, precision=prec)
o = ol.Oracle(seq.cam, seq.par, N + 8)
e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
names = ["r", "q", "v", "w"]
sl = [slice(0, 3), slice(3, 7), slice(7, 10), slice(10, 13)]
for t, (kps, desc) in enumerate(seq.frames):
    ie = e.step(kps, desc)
    io = o.step(kps, desc, ol.ALGORITHMIC)
    x, fp, P = e.get_state()
    xo, fpo, Po = o.x13(), o.feature_pos(), o.P()
    blk = " ".join(f"{n}:{np.abs(x[s]-xo[s]).max()/max(np.abs(xo[s]).max(),1e-300):.1e}" for n, s in zip(names, sl))
    fe = np.abs(fp - fpo)
    print(f"t={t} cnt e=({ie.n_matches},{ie.n_inliers},{ie.n_rescued},{ie.n_hypotheses}) o=({io.n_matches},{io.n_inliers},{io.n_rescued},{io.n_hypotheses}) "
          f"| {blk} | feat abs max xyz {fe[:, :3].max():.1e} ang {fe[:, 3:5].max():.1e} rho {fe[:, 5].max():.1e} (rho rel {(fe[:,5]/np.abs(fpo[:,5])).max():.1e})"
          f" | P fro {np.linalg.norm(P-Po)/np.linalg.norm(Po):.2e} max {np.abs(P-Po).max()/np.abs(Po).max():.2e} diag rel {np.abs(np.diag(P)-np.diag(Po)).max()/np.abs(np.diag(Po)).max():.2e}")
