"""N = 5000, fp32 covariance, three frames against tests/golden/oracle_n5000_f3_summary.npz: block errors per frame for the two ways
of forming B = inv(L) H P (arg 1: 0 auto = inverse + GEMM above 2048 rows, 1 always inside the sweep) and for the fp64 engine
(arg 2: precision 1 fp32 / 0 fp64).  usage: diag_n5000_paths.py [path] [precision]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from openekfmonoslam_amd import engine  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402
from parity_metric import block_errs  # noqa: E402

path = int(sys.argv[1]) if len(sys.argv) > 1 else 0
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
z = np.load(os.path.join(ROOT, "tests", "golden", "oracle_n5000_f3_summary.npz"))
N, F = int(z["n_features"]), int(z["frames"])
seq = SyntheticSequence(N, F, width=int(z["width"]), height=int(z["height"]))
e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=prec)
e.set_update_path(path)
e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
idx = z["sample_idx"]
for t in range(F):
    i = e.step(*seq.frames[t])
    same = [i.n_predicted, i.n_matches, i.n_hypotheses, i.n_inliers, i.n_outliers, i.n_rescued, i.status] == list(z["info"][t])
    x, fp, P = e.get_state()
    be = block_errs(x, fp, z[f"x13_t{t}"], z[f"feature_pos_t{t}"])
    maxabs = float(z[f"maxabs_t{t}"])
    be["P_sample_max"] = float(np.abs(P[np.ix_(idx, idx)] - z[f"sample_t{t}"]).max() / maxabs)
    be["P_diag_max"] = float(np.abs(np.diag(P) - z[f"diag_t{t}"]).max() / maxabs)
    d = fp[:, 5] - z[f"feature_pos_t{t}"][:, 5]
    j = int(np.argmax(np.abs(d)))
    dg = z[f"diag_t{t}"]
    pct = np.percentile(np.abs(d), [50, 90, 99, 99.9, 100])
    var_rho = dg[13 + 6 * np.arange(N) + 5]
    worst = np.argsort(-np.abs(d))[:5]
    print("   |rho err| percentiles 50/90/99/99.9/100:", " ".join(f"{v:.2e}" for v in pct),
          "; worst five (feature, err, rho variance, engine variance):",
          [(int(k), f"{d[k]:.2e}", f"{var_rho[k]:.2e}", f"{P[13 + 6 * k + 5, 13 + 6 * k + 5]:.2e}") for k in worst],
          f"; median rho variance {np.median(var_rho):.2e}", flush=True)
    del P
    print(f"path {path} precision {prec} frame {t}: decisions same {same}", {k: f"{v:.2e}" for k, v in be.items()},
          f"worst rho: feature {j} value {z[f'feature_pos_t{t}'][j, 5]:.4e} err {d[j]:.2e}, max|rho| {np.abs(z[f'feature_pos_t{t}'][:, 5]).max():.3f}", flush=True)
e.close()
