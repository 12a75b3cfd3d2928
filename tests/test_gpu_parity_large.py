"""GPU parity at the map sizes the numbers are quoted on (BASELINE configs[2..4]): the HIP engine with fp32 STORAGE of the
covariance against the fp64 oracle (ALGORITHMIC variant: block-sparse H, Cholesky, P -= B'B -- Update.cpp:282-319 restated
without the dense n x n temporaries; validated against the LITERAL variant at N <= 200 in tests/test_oracle_selfcheck.py).

Tolerance: the north-star 1e-5 and nothing wider, at every size: per block -- camera r, q, v, w and the feature anchors, theta,
phi, rho (max-norm of the difference over the block's max-norm), P in max-norm and in Frobenius norm --, for every single
feature parameter against max(|own value|, 1e-4), and identical decision counters (predicted / matches / hypotheses / inliers /
rescued) on every frame.  The configuration held to it is EKF_PRECISION_AUTO (precision 4): the exact update -- B = inv(L) H P in
fp64, the rank-m downdate accumulated exactly on the int8 MFMA (the reference computes in double, Core/Base.h:67,
Update.cpp:105-108, 214-218) -- on an fp32-stored covariance up to 1024 features (EKF_PRECISION_F32_EXACT, precision 2: one rounding
to fp32 per entry and update) and on an fp64-stored one above (EKF_PRECISION_F64_EXACT, precision 3): fp32 storage alone leaves 1e-5
component-wise on fresh maps of >= 1400 features (3.05e-5 at N = 1400 frame 2, 3.6e-4 at N = 5000 frame 3, round 4).  There is no
widened gate and no measured "floor" any more.  The fast configuration EKF_PRECISION_F32 (precision 1: fp32 B, fp32 MFMA
accumulation) is run on the same frames with its decisions and its BLOCK errors asserted where it meets them (N <= 2000 and
the first N = 5000 frame) and its component-wise figure printed: it is not the parity configuration (measured up to 7.6e-5
there; N = 5000 frames 2-3: inverse-depth block 1.9e-5 / 4.5e-5, profiles/r03_n5000_three_frames_fp32.txt).
"""
import os

import numpy as np
import pytest

from openekfmonoslam_amd.synth import SyntheticSequence
from parity_metric import F32_TOL, block_errs, over_tolerance, parity_report

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXACT, FAST, AUTO = 2, 1, 4  # EKF_PRECISION_F32_EXACT, EKF_PRECISION_F32 (fast, block-wise only), EKF_PRECISION_AUTO (the parity configuration at any size)
COUNTERS = ("n_predicted", "n_matches", "n_hypotheses", "n_inliers", "n_outliers", "n_rescued", "status")


@pytest.fixture(scope="module")
def eng_mod():
    from openekfmonoslam_amd import engine

    lib = engine.load_library()
    assert lib.ekf_device_count() >= 1, "no MI355X visible"
    return engine


def run_pair(eng_mod, ol, seq, frames, precision=EXACT, path=0, key=None, P0=None):
    """`key`: the oracle's frames are shared with the other tests that pass the same key (same sequence, same start `P0`)"""
    N = seq.n_features
    P0 = seq.P0 if P0 is None else P0
    e = eng_mod.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=precision)
    e.set_update_path(path)
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
    if key is None:
        o = ol.Oracle(seq.cam, seq.par, N + 8)
        o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
    else:
        ref = ol.oracle_frames(seq, frames, P0, key)
    worst = {}
    for t in range(frames):
        ie = e.step(*seq.frames[t])
        if key is None:
            io = o.step(*seq.frames[t], ol.ALGORITHMIC)
            xo, fpo, Po = o.x13(), o.feature_pos(), o.P()
        else:
            io, xo, fpo, Po = ref[t]
        for f in COUNTERS:
            assert getattr(ie, f) == getattr(io, f), (t, f, getattr(ie, f), getattr(io, f))
        x, fp, P = e.get_state()
        be = parity_report(x, fp, P, xo, fpo, Po)
        for k, v in be.items():
            worst[k] = max(worst.get(k, 0.0), v)
        bad = over_tolerance(be, F32_TOL, N, componentwise=precision != FAST)
        assert not bad, f"frame {t}: blocks over {F32_TOL:g}: {bad}  (all: {be})"
    e.close()
    return worst


BOTH = pytest.mark.parametrize("precision", [EXACT, FAST], ids=["exact", "fast"])


# the scene of the round-1 driver run (25 frames asked for), the round-1 builder runs (70) and the current generator's
# (fixed 100-frame horizon), plus two more seeds
@BOTH  # (the top decorator varies fastest: the two precisions of a scene are adjacent and share the oracle's frames)
@pytest.mark.parametrize("kw", [dict(horizon=25), dict(horizon=70), dict(), dict(seed=0xC0FFEE), dict(seed=20260102)],
                         ids=["scene25", "scene70", "scene100", "seedA", "seedB"])
def test_n1000_fp32_four_frames_vs_oracle(eng_mod, oracle_lib, kw, precision):
    """BASELINE configs[2]: N = 1000, fp32 covariance, four frames: every block and (exact configuration) every feature
    parameter <= 1e-5."""
    seq = SyntheticSequence(1000, 4, **kw)
    worst = run_pair(eng_mod, oracle_lib, seq, 4, precision=precision, key=("n1000_4f", tuple(sorted(kw.items()))))
    print(f"N=1000 precision {precision} worst errors over 4 frames:", {k: f"{v:.2e}" for k, v in worst.items()})


@BOTH
def test_n1000_fp32_inverse_and_gemm_path_vs_oracle(eng_mod, oracle_lib, precision):
    """The same bar with B = inv(L) H P forced onto the explicit inverse + GEMM (the path of updates with more than 2048
    measurement rows; at N = 1000 the sweep path is the default and the five scenes above run it)."""
    seq = SyntheticSequence(1000, 3)
    worst = run_pair(eng_mod, oracle_lib, seq, 3, precision=precision, path=2, key="n1000_3f")
    print(f"N=1000 precision {precision} (inverse + GEMM) worst errors over 3 frames:", {k: f"{v:.2e}" for k, v in worst.items()})


@BOTH
def test_n2000_fp32_two_frames_vs_oracle(eng_mod, oracle_lib, precision):
    """BASELINE configs[3] map size (N = 2000, 1280x720), fp32 covariance, unsharded engine, two frames."""
    seq = SyntheticSequence(2000, 2, width=1280, height=720)
    # (the symmetrised start the sharded suite uses: ONE oracle run of this scene, a minute of host time, serves both;
    # the explicit-average downdate after an asymmetric upload is exercised at N = 1000 and 1400)
    worst = run_pair(eng_mod, oracle_lib, seq, 2, precision=precision, key="n2000_1280x720_2f_sym", P0=0.5 * (seq.P0 + seq.P0.T))
    print(f"N=2000 precision {precision} worst errors over 2 frames:", {k: f"{v:.2e}" for k, v in worst.items()})


@pytest.mark.parametrize("precision", [AUTO, FAST], ids=["auto", "fast"])
def test_n1400_two_frames_vs_oracle(eng_mod, oracle_lib, precision):
    """N = 1400 (n_pad = 8448): a map on which fp32 storage of the covariance alone flips a matching decision in the second frame
    and leaves 3e-5 component-wise (round 4).  EKF_PRECISION_AUTO stores such a map in fp64 (EKF_PRECISION_F64_EXACT: same exact
    int8 update): every block and every feature parameter within the plain 1e-5 over two frames, decisions identical."""
    seq = SyntheticSequence(1400, 2, width=1280, height=720)
    w = run_pair(eng_mod, oracle_lib, seq, 2, precision=precision, key="n1400_1280x720_2f")
    print(f"N=1400 precision {precision} worst errors:", {k: f"{v:.2e}" for k, v in w.items()})


@BOTH
def test_n5000_fp32_against_committed_summary(eng_mod, precision):
    """BASELINE configs[4] map size (N = 5000, 1920x1080, fp32) on ONE GPU against the committed oracle summary
    (tests/golden/make_large_fixture.py; the oracle needs ~10 minutes per frame at this size, so it is not run here)."""
    path = os.path.join(GOLDEN, "oracle_n5000_f1_summary.npz")
    if not os.path.exists(path):
        pytest.skip("summary fixture not minted")
    z = np.load(path)
    N, F = int(z["n_features"]), int(z["frames"])
    seq = SyntheticSequence(N, F, width=int(z["width"]), height=int(z["height"]))
    # the generator is part of the contract: same inputs as when the summary was minted
    assert np.isclose(np.trace(seq.P0), float(z["input_P0_trace"]), rtol=1e-13, atol=0)
    assert np.isclose(seq.frames[0][0]["x"].astype(np.float64).sum(), float(z["input_kps0_sum"]), rtol=1e-13, atol=0)
    e = eng_mod.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=precision)
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    for t in range(F):
        i = e.step(*seq.frames[t])
        assert [i.n_predicted, i.n_matches, i.n_hypotheses, i.n_inliers, i.n_outliers, i.n_rescued, i.status] == list(z["info"][t])
    x, fp, P = e.get_state()
    e.close()
    be = block_errs(x, fp, z["x13"], z["feature_pos"])
    maxabs = float(z["maxabs"])
    idx = z["sample_idx"]
    be["P13_max"] = float(np.abs(P[:13, :13] - z["P13"]).max() / maxabs)
    # the camera block against its OWN largest entry (1e3-1e5 times smaller than max|P|): a stricter reading than the
    # norm-wise measure of P, held to 1e-4 (measured 1.05e-5 after the m = 9462 update of this frame)
    p13_own = float(np.abs(P[:13, :13] - z["P13"]).max() / np.abs(z["P13"]).max())
    be["P_sample_max"] = float(np.abs(P[np.ix_(idx, idx)] - z["sample"]).max() / maxabs)
    be["P_diag_max"] = float(np.abs(np.diag(P) - z["diag"]).max() / maxabs)
    be["trace"] = abs(float(np.trace(P)) - float(z["trace"])) / float(z["trace"])
    be["fro"] = abs(float(np.linalg.norm(P)) - float(z["fro"])) / float(z["fro"])
    print(f"N=5000 precision {precision} vs committed oracle summary:", {k: f"{v:.2e}" for k, v in be.items()})
    print("camera block relative to its own max:", f"{p13_own:.2e}")
    bad = {k: v for k, v in be.items() if (k != "features_componentwise" or precision != FAST) and not v <= F32_TOL}
    assert not bad, bad
    assert p13_own <= 1e-4, p13_own


@pytest.mark.parametrize("precision,expect,tol", [(AUTO, 0, 1e-9), (3, 3, None), (EXACT, EXACT, "survey")],
                         ids=["auto", "fp64_stored_exact", "fp32_stored_exact"])
def test_n5000_three_frames_against_committed_summary(eng_mod, precision, expect, tol):
    """configs[4] map size over THREE frames: after every frame the engine's decisions, state blocks, EVERY feature parameter, camera
    block, diagonal, trace, Frobenius norm and a 64 x 64 sample of P against the per-frame oracle summaries of
    tests/golden/oracle_n5000_f3_summary.npz (77 minutes of oracle time, minted once by make_large_fixture.py 5000 3).
    EKF_PRECISION_AUTO -- the parity configuration at any size -- is all-fp64 at this capacity: everything within 1e-9 (measured
    1e-12).  The exact int8 update on an fp64-stored covariance (EKF_PRECISION_F64_EXACT, 2.4 x faster) holds every BLOCK at 1e-5 on
    all three frames and every feature parameter on the first two (2.8e-6); on the third one far feature's inverse depth is 2.5e-5 ...
    3.4e-5 of its own value off (fp32 storage, EKF_PRECISION_F32_EXACT: 3.6e-4, round 4) -- asserted block-wise, the component-wise
    figure of frame 3 held to 1e-4 as a regression guard: it is why AUTO does not select it above 2048 features.
    fp32_stored_exact -- configs[4] says "fp32": EKF_PRECISION_F32_EXACT over the same three frames (what `bench.py --gpus 8` runs per
    rank) under the gates SURVEY 8(d) states for every reported number: identical decisions on every frame, every state BLOCK within
    1e-5, P within 1e-5 in max-norm (camera block, diagonal, 64 x 64 sample) and in Frobenius norm.  Its component-wise figure (one far
    feature's inverse depth against its own value: 1.7e-5 / 1.4e-4 on frames 2 / 3, round 4) is printed, and held to 1e-3 as a
    regression guard only -- the configuration that holds 1e-5 component-wise at this size is AUTO."""
    path = os.path.join(GOLDEN, "oracle_n5000_f3_summary.npz")
    if not os.path.exists(path):
        pytest.skip("summary fixture not minted")
    z = np.load(path)
    N, F = int(z["n_features"]), int(z["frames"])
    assert F == 3
    seq = SyntheticSequence(N, F, width=int(z["width"]), height=int(z["height"]))
    assert np.isclose(np.trace(seq.P0), float(z["input_P0_trace"]), rtol=1e-13, atol=0)
    assert np.isclose(seq.frames[0][0]["x"].astype(np.float64).sum(), float(z["input_kps0_sum"]), rtol=1e-13, atol=0)
    e = eng_mod.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=precision)
    assert e.precision == expect
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    idx = z["sample_idx"]
    reports = []
    for t in range(F):
        i = e.step(*seq.frames[t])
        assert [i.n_predicted, i.n_matches, i.n_hypotheses, i.n_inliers, i.n_outliers, i.n_rescued, i.status] == list(z["info"][t]), t
        if tol == 1e-9 and t < F - 1:
            continue  # (the all-fp64 engine: the last frame tells; every get_state moves 7 GB)
        x, fp, P = e.get_state()
        be = block_errs(x, fp, z[f"x13_t{t}"], z[f"feature_pos_t{t}"])
        maxabs = float(z[f"maxabs_t{t}"])
        be["P13_max"] = float(np.abs(P[:13, :13] - z[f"P13_t{t}"]).max() / maxabs)
        p13_own = float(np.abs(P[:13, :13] - z[f"P13_t{t}"]).max() / np.abs(z[f"P13_t{t}"]).max())
        be["P_sample_max"] = float(np.abs(P[np.ix_(idx, idx)] - z[f"sample_t{t}"]).max() / maxabs)
        be["P_diag_max"] = float(np.abs(np.diag(P) - z[f"diag_t{t}"]).max() / maxabs)
        be["trace"] = abs(float(np.trace(P)) - float(z[f"trace_t{t}"])) / float(z[f"trace_t{t}"])
        be["fro"] = abs(float(np.linalg.norm(P)) - float(z[f"fro_t{t}"])) / float(z[f"fro_t{t}"])
        del P
        print(f"N=5000 precision {e.precision} frame {t} vs committed oracle summary:", {k: f"{v:.2e}" for k, v in be.items()},
              f"camera block vs its own max {p13_own:.2e}")
        reports.append((t, be, p13_own))
    e.close()
    for t, be, p13_own in reports:
        if tol == "survey":
            bad = {k: v for k, v in be.items() if k != "features_componentwise" and not v <= F32_TOL}
            if not be["features_componentwise"] <= 1e-3:
                bad["features_componentwise"] = be["features_componentwise"]
        elif tol is not None:
            bad = {k: v for k, v in be.items() if not v <= tol}
        else:
            bad = {k: v for k, v in be.items() if k != "features_componentwise" and not v <= F32_TOL}
            if not be["features_componentwise"] <= (F32_TOL if t < 2 else 1e-4):
                bad["features_componentwise"] = be["features_componentwise"]
        assert not bad, (t, bad)
        assert p13_own <= 1e-4, (t, p13_own)
