"""Error measures shared by the GPU parity tests and bench.py's parity gate: the HIP engine against the
fp64 oracle.  TEST INFRASTRUCTURE (numpy only)."""
import numpy as np

F64_TOL = 1e-9   # fp64 engine vs fp64 oracle (different summation orders / FMA only)
F32_TOL = 1e-5   # the north-star tolerance for the fp32-covariance configuration
# Every gate below is the north-star figure, UN-widened: each block AND every single feature parameter (against
# max(|own value|, 1e-4): SURVEY.md 8(d), "abs floor 1e-9").  The configuration that meets it is EKF_PRECISION_AUTO: the exact
# update (fp64 B, exact int8-digit downdate) on an fp32-stored covariance up to 1024 features (EKF_PRECISION_F32_EXACT: worst
# component 2.5e-6 at N = 1000 over five scenes x four frames) and on an fp64-stored one above (EKF_PRECISION_F64_EXACT; fp32 storage
# alone leaves 1e-5 on fresh maps of >= 1400 features).  The fast fp32 MFMA configuration (EKF_PRECISION_F32) does NOT meet the
# component-wise reading at N >= 1000 (measured up to 7.6e-5; cause: profiles/r03_parity_attribution.txt): callers that run it pass
# componentwise=False and get the figure held to FAST_COMPONENT_CEILING only -- a REGRESSION guard of a shipped secondary
# configuration, not a parity claim.
FAST_COMPONENT_CEILING = 2e-4
ASSERTED_BLOCKS = ("r", "q", "v", "w", "feat_xyz", "feat_theta", "feat_phi", "feat_rho", "P_max", "P_fro")


def rel_max(a, b):
    """max |a-b| / max |b| : the norm-wise measure SURVEY.md 8(d) prescribes for P."""
    b = np.asarray(b)
    return float(np.abs(np.asarray(a) - b).max() / max(np.abs(b).max(), 1e-300))


def rel_fro(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def state_err(x, fp, xo, fpo):
    """max relative error over the camera 13-vector and the feature blocks; components smaller than 1e-4 are
    measured against 1e-4 (an absolute floor of 1e-9 at the 1e-5 tolerance)."""
    a = np.concatenate([x, fp.reshape(-1)])
    b = np.concatenate([xo, fpo.reshape(-1)])
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-4)).max())


def block_errs(x, fp, xo, fpo):
    """Relative error of every physically homogeneous block of the state, max-norm of the difference over the max-norm
    of the block ("within 1e-5 rel of the reference" for a vector quantity): camera r, q, v, w; feature anchors /
    XYZ points (components 0..2), theta, phi, rho (components 3, 4, 5 of the inverse-depth features).
    `features_blockwise` is the worst of the four feature blocks (rounds 1-2 reported it as `features`).
    `features_componentwise` is the strictest reading (abs floor 1e-9 at 1e-5): every component against its
    OWN magnitude with a floor of 1e-4 -- an angle that happens to lie at 7e-5 rad or an inverse depth estimated at 4e-3 is
    then held to an absolute 1e-9 ... 4e-8.  Gated by over_tolerance() at `tol` like every block."""
    out = {}
    for name, sl in (("r", slice(0, 3)), ("q", slice(3, 7)), ("v", slice(7, 10)), ("w", slice(10, 13))):
        out[name] = float(np.abs(x[sl] - xo[sl]).max() / max(np.abs(xo[sl]).max(), 1e-9))
    fp, fpo = np.asarray(fp).reshape(-1, 6), np.asarray(fpo).reshape(-1, 6)
    worst = 0.0
    for name, sl in (("feat_xyz", slice(0, 3)), ("feat_theta", slice(3, 4)), ("feat_phi", slice(4, 5)), ("feat_rho", slice(5, 6))):
        scale = np.abs(fpo[:, sl]).max() if fpo.size else 0.0
        out[name] = float(np.abs(fp[:, sl] - fpo[:, sl]).max() / scale) if scale > 0 else float(np.abs(fp[:, sl] - fpo[:, sl]).max() if fp.size else 0.0)
        worst = max(worst, out[name])
    out["features_blockwise"] = worst
    out["features_componentwise"] = float((np.abs(fp - fpo) / np.maximum(np.abs(fpo), 1e-4)).max()) if fp.size else 0.0
    return out


def parity_report(x, fp, P, xo, fpo, Po):
    """All block errors of one comparison (state blocks + P in max-norm and Frobenius norm)."""
    be = block_errs(x, fp, xo, fpo)
    be["P_max"], be["P_fro"] = rel_max(P, Po), rel_fro(P, Po)
    return be


def over_tolerance(be, tol, n_features=0, componentwise=True):
    """Blocks (and, unless componentwise=False, the worst single feature parameter) above `tol`; with componentwise=False (the
    fast fp32 configuration) the worst feature parameter is held to FAST_COMPONENT_CEILING, a regression guard."""
    bad = {k: v for k, v in be.items() if k in ASSERTED_BLOCKS and not v <= tol}
    if componentwise and not be.get("features_componentwise", 0.0) <= tol:
        bad["features_componentwise"] = be["features_componentwise"]
    if not componentwise and not be.get("features_componentwise", 0.0) <= max(tol, FAST_COMPONENT_CEILING):
        bad["features_componentwise (regression ceiling of the fast configuration)"] = be["features_componentwise"]
    return bad
