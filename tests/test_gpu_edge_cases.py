"""Edge cases of the path, HIP engine through the C ABI vs the oracle: empty and ragged inputs (no keypoints, no
visible features, empty map), update sizes that straddle every blocking boundary of the solver (32-wide panels,
64/128-wide GEMM tiles, the 256 threshold between the small and the MFMA inverse levels), a frame in which every
match is an outlier, an indefinite innovation covariance, and the fp32-covariance drift over a long run."""
import numpy as np
import pytest

from openekfmonoslam_amd.ekftypes import KEYPOINT_DTYPE, MATCH_DTYPE
from openekfmonoslam_amd.synth import SyntheticSequence
from tests.oracle_lib import ALGORITHMIC, align_to_matches
from tests.test_gpu_parity import (F32_TOL, F32_TOL_OMEGA, F64_TOL, assert_state_close, block_errs, eng_mod, make_pair,  # noqa: F401
                                   rel_fro, rel_max)

pytestmark = pytest.mark.gpu
INFO = ("n_predicted", "n_matches", "n_hypotheses", "n_inliers", "n_outliers", "n_rescued", "status")


def _same_info(gi, oi, where):
    for f in INFO:
        assert getattr(gi, f) == getattr(oi, f), (where, f, getattr(gi, f), getattr(oi, f))


def _matches_from_predictions(preds, M):
    """M matches: the first M predicted features, measured at their prediction plus a fraction of a pixel"""
    assert len(preds) >= M
    rng = np.random.default_rng(100 + M)
    m = np.zeros(M, dtype=MATCH_DTYPE)
    m["featureIndex"] = preds["featureIndex"][:M]
    m["keypointIndex"] = np.arange(M)
    m["imagePos"] = preds["imagePos"][:M] + rng.normal(0.0, 0.3, (M, 2))
    return m


def test_frame_without_keypoints(eng_mod, oracle_lib, seq12):
    """no detections: prediction only, nothing matched, no update (EKF.cpp:337-430 with empty vectors)"""
    e, o = make_pair(eng_mod, oracle_lib, seq12)
    kps = np.zeros(0, dtype=KEYPOINT_DTYPE)
    desc = np.zeros((0, 32), dtype=np.uint8)
    gi = e.step(kps, desc)
    oi = o.step(kps, desc, ALGORITHMIC)
    _same_info(gi, oi, "empty frame")
    assert gi.n_matches == 0 and gi.n_predicted == 12
    assert_state_close(e, o, 1e-12, "empty frame")
    # and a normal frame afterwards
    gi = e.step(*seq12.frames[1])
    oi = o.step(*seq12.frames[1], ALGORITHMIC)
    _same_info(gi, oi, "frame after the empty one")
    assert_state_close(e, o, F64_TOL, "frame after the empty one")


def test_no_feature_visible(eng_mod, oracle_lib, seq12):
    """camera turned away: every feature fails the visibility test (MeasurementPrediction.cpp:233-262)"""
    e, o = make_pair(eng_mod, oracle_lib, seq12)
    x = np.array(seq12.x13)
    x[3:7] = [0.0, 0.0, 1.0, 0.0]  # half a turn about y: the map is behind the camera
    for f in (e, o):
        f.set_state(x, seq12.feature_pos, seq12.feature_type, seq12.feature_desc, seq12.P0)
    gi = e.step(*seq12.frames[0])
    oi = o.step(*seq12.frames[0], ALGORITHMIC)
    _same_info(gi, oi, "nothing visible")
    assert gi.n_predicted == 0 and gi.n_matches == 0
    assert_state_close(e, o, 1e-12, "nothing visible")
    assert len(e.unseen_features()) == 12


def test_empty_map(eng_mod, oracle_lib, seq12):
    """N = 0: the 13-state filter alone (initState / initCovariance, then steps)"""
    e = eng_mod.EkfEngine(seq12.cam, seq12.par, 16)
    o = oracle_lib.Oracle(seq12.cam, seq12.par, 16)
    e.reset()
    o.reset()
    for t in range(2):
        gi = e.step(*seq12.frames[t])
        oi = o.step(*seq12.frames[t], ALGORITHMIC)
        _same_info(gi, oi, f"empty map, frame {t}")
    assert e.n == 13
    assert_state_close(e, o, 1e-12, "empty map")


@pytest.mark.parametrize("path", [pytest.param(1, id="sweep"), pytest.param(2, id="gemm")])
@pytest.mark.parametrize("M", [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 160])
def test_update_sizes_across_block_boundaries(eng_mod, oracle_lib, M, path):
    """m = 2M rows of S: 2 .. 320, i.e. below / at / above one panel, one fp64 GEMM tile (64), one fp32 tile (128), and
    the 256 switch of the inverse's doubling levels; on both ways of forming B = inv(L) H P (ekf_set_update_path)"""
    seq = SyntheticSequence(170, 1, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    e, o = make_pair(eng_mod, oracle_lib, seq)
    e.set_update_path(path)
    e.predict()
    o.predict()
    e.predict_measurements()
    preds, Hs, Hf = o.predict_measurements()
    mo = _matches_from_predictions(preds, M)
    mp, mHs, mHf = align_to_matches(preds, Hs, Hf, mo)
    assert o.update(mo, mp, mHs, mHf, ALGORITHMIC) == 0
    e.update(mo)
    assert_state_close(e, o, F64_TOL, f"update with M = {M}")


@pytest.mark.parametrize("path", [pytest.param(1, id="sweep"), pytest.param(2, id="gemm")])
@pytest.mark.parametrize("M", [1, 16, 17, 32, 33, 48, 49, 64, 65, 80, 96, 97, 128, 129, 160])
def test_update_sizes_two_panels_per_launch(eng_mod, oracle_lib, M, path):
    """the same sizes with the Cholesky sweep in its two-panels-per-launch form (ekf_set_sweep_mode(EKF_SWEEP_PAIRS),
    csrc/chol_pair.h): odd and even panel counts, a short last panel in either half of a pair, both ways of forming B"""
    seq = SyntheticSequence(170, 1, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    e, o = make_pair(eng_mod, oracle_lib, seq)
    e.set_update_path(path)
    e.set_sweep_mode(0)
    e.predict()
    o.predict()
    e.predict_measurements()
    preds, Hs, Hf = o.predict_measurements()
    mo = _matches_from_predictions(preds, M)
    mp, mHs, mHf = align_to_matches(preds, Hs, Hf, mo)
    assert o.update(mo, mp, mHs, mHf, ALGORITHMIC) == 0
    e.update(mo)
    assert_state_close(e, o, F64_TOL, f"update with M = {M}, two panels per launch")


@pytest.mark.parametrize("precision", [pytest.param(1, id="fp32"), pytest.param(0, id="fp64")])
def test_sweep_modes_agree_over_frames(eng_mod, precision):
    """N = 1000, fp32 covariance, four frames: one panel per launch against two panels per launch and against the default
    (by size: pairs once the rows of B are the longest role) -- identical decisions, states equal to rounding (the 64 x 64
    look-ahead inverse is the same algebra in another order)"""
    seq = SyntheticSequence(1000, 4)
    out = []
    for mode in (1, 0, 2):
        e = eng_mod.EkfEngine(seq.cam, seq.par, 1000, max_keypoints=len(seq.frames[0][0]) + 64, precision=precision)
        e.set_sweep_mode(mode)
        e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
        infos = [e.step(*seq.frames[t]) for t in range(4)]
        out.append((infos, e.get_state()))
        e.close()
    for other in out[1:]:
        for a, b in zip(out[0][0], other[0]):
            assert (a.n_matches, a.n_hypotheses, a.n_inliers, a.n_rescued) == (b.n_matches, b.n_hypotheses, b.n_inliers, b.n_rescued)
        (xa, fa, Pa), (xb, fb, Pb) = out[0][1], other[1]
        tol = 1.0 if precision == 1 else 1e-6  # fp64: the two launch schemes agree to 1e-12
        assert rel_max(Pb, Pa) <= 1e-6 * tol and np.abs(xa - xb).max() <= 1e-7 * tol and np.abs(fa - fb).max() <= 1e-6 * tol


def test_update_sizes_fp32(eng_mod, oracle_lib):
    seq = SyntheticSequence(170, 1, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    for M, path in ((17, 1), (64, 1), (129, 1), (160, 1), (17, 2), (129, 2), (160, 2)):
        e, o = make_pair(eng_mod, oracle_lib, seq, precision=1)
        e.set_update_path(path)
        e.predict()
        o.predict()
        e.predict_measurements()
        preds, Hs, Hf = o.predict_measurements()
        mo = _matches_from_predictions(preds, M)
        mp, mHs, mHf = align_to_matches(preds, Hs, Hf, mo)
        assert o.update(mo, mp, mHs, mHf, ALGORITHMIC) == 0
        e.update(mo)
        x, fp, P = e.get_state()
        assert rel_max(P, o.P()) <= F32_TOL and rel_fro(P, o.P()) <= F32_TOL, M


def test_all_matches_are_outliers(eng_mod, oracle_lib, seq12):
    """every keypoint displaced by the same gross offset: RANSAC keeps its one-point hypothesis, the rest go through
    the re-prediction / rescue path (EKF.cpp:441-556)"""
    e, o = make_pair(eng_mod, oracle_lib, seq12)
    kps, desc = seq12.frames[0]
    bad = kps.copy()
    rng = np.random.default_rng(9)
    bad["x"] += rng.uniform(-6, 6, len(bad)).astype(np.float32)
    bad["y"] += rng.uniform(-6, 6, len(bad)).astype(np.float32)
    gi = e.step(bad, desc)
    oi = o.step(bad, desc, ALGORITHMIC)
    _same_info(gi, oi, "scattered keypoints")
    assert_state_close(e, o, F64_TOL, "scattered keypoints")


def test_indefinite_innovation_covariance_is_reported(eng_mod, seq12):
    """S = H P H' + R not positive definite (P made indefinite on purpose): the engine reports it
    (EKF_ERR_NOT_POSITIVE_DEFINITE) instead of returning cv::invert's silent zeros (Update.cpp:101-103)"""
    e = eng_mod.EkfEngine(seq12.cam, seq12.par, 16, max_keypoints=128)
    P = -1e3 * np.eye(seq12.state_dim)
    e.set_state(seq12.x13, seq12.feature_pos, seq12.feature_type, seq12.feature_desc, P)
    e.predict()
    preds, _, _ = e.predict_measurements()
    m = np.zeros(4, dtype=MATCH_DTYPE)
    m["featureIndex"] = preds["featureIndex"][:4]
    m["imagePos"] = preds["imagePos"][:4] + 0.5
    with pytest.raises(eng_mod.EkfError) as ei:
        e.update(m)
    assert ei.value.code == 3


def test_fp32_covariance_drift_over_90_frames(eng_mod, oracle_lib):
    """'for fp32 configs also report the drift after 90 frames' -- N = 100, fp32 covariance vs the fp64
    oracle, same frames; decisions (matches / inliers / rescued) must stay identical, errors inside the stated bars"""
    seq = SyntheticSequence(100, 90)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=1)
    for t in range(90):
        gi = e.step(*seq.frames[t])
        oi = o.step(*seq.frames[t], ALGORITHMIC)
        _same_info(gi, oi, f"frame {t}")
    x, fp, P = e.get_state()
    be = block_errs(x, fp, o.x13(), o.feature_pos())
    pm, pf = rel_max(P, o.P()), rel_fro(P, o.P())
    print(f"fp32 drift after 90 frames: P max {pm:.2e} fro {pf:.2e} blocks {be}")
    assert pm <= F32_TOL and pf <= F32_TOL
    for name in ("r", "q", "v", "features_blockwise"):
        assert be[name] <= F32_TOL, (name, be[name])
    assert be["w"] <= F32_TOL_OMEGA


def test_abi_error_paths(eng_mod, seq12):
    """capacity and argument errors come back as status codes (EkfError), the engine stays usable afterwards"""
    e = eng_mod.EkfEngine(seq12.cam, seq12.par, 12, max_keypoints=16)
    with pytest.raises(eng_mod.EkfError) as ei:   # more features than the engine was created for
        big = SyntheticSequence(13, 1)
        e.set_state(big.x13, big.feature_pos, big.feature_type, big.feature_desc, big.P0)
    assert ei.value.code == 2
    e.set_state(seq12.x13, seq12.feature_pos, seq12.feature_type, seq12.feature_desc, seq12.P0)
    with pytest.raises(eng_mod.EkfError) as ei:   # more keypoints than max_keypoints
        e.step(*seq12.frames[0])
    assert ei.value.code == 2
    with pytest.raises(eng_mod.EkfError) as ei:   # map is full
        e.add_features(np.array([[100.0, 100.0]]))
    assert ei.value.code == 2
    with pytest.raises(eng_mod.EkfError) as ei:   # feature index out of range in a match
        m = np.zeros(1, dtype=MATCH_DTYPE)
        m["featureIndex"] = 99
        e.predict()
        e.predict_measurements()
        e.update(m)
    assert ei.value.code == 1
    with pytest.raises(eng_mod.EkfError):          # removing a feature that does not exist
        e.remove_features([40])
    with pytest.raises(eng_mod.EkfError) as ei:   # an update path that does not exist
        e.set_update_path(3)
    assert ei.value.code == 1
    with pytest.raises(eng_mod.EkfError) as ei:   # a sweep launch scheme that does not exist
        e.set_sweep_mode(5)
    assert ei.value.code == 1
    # still healthy: a normal prediction + update on a subset of the keypoints
    kps, desc = seq12.frames[0]
    info = e.step(kps[:16], desc[:16])
    assert info.status == 0


def test_remove_every_feature_then_regrow(eng_mod, oracle_lib, seq12):
    e, o = make_pair(eng_mod, oracle_lib, seq12)
    idx = np.arange(12, dtype=np.int32)
    e.remove_features(idx)
    o.remove_features(idx)
    assert e.N == 0 and e.n == 13
    assert_state_close(e, o, 1e-12, "all features removed")
    uv = np.array([[100.5, 80.25], [400.0, 300.0], [320.0, 240.0]])
    e.add_features(uv)
    for p in uv:
        o.add_feature(p)
    assert e.N == 3 and e.n == 31
    assert_state_close(e, o, F64_TOL, "regrown map")
    gi = e.step(*seq12.frames[0])
    oi = o.step(*seq12.frames[0], ALGORITHMIC)
    _same_info(gi, oi, "step on the regrown map")
    assert_state_close(e, o, F64_TOL, "step on the regrown map")


def test_concurrent_engines_give_the_sequential_result(eng_mod):
    """Two filters stepped from two threads at once (their kernels interleave on the GPU) must each reproduce, bit for
    bit, what the same filter computes alone.  Guards the in-launch data flow of the sweep: the right-hand-side blocks
    of k_chol_step once overwrote rows other blocks of the same launch were still reading, which only showed under
    contention (N = 1000: up to four right-hand-side blocks per launch; with the defect this test fails)."""
    import threading

    seq = SyntheticSequence(1000, 4)

    def run(out, k):
        e = eng_mod.EkfEngine(seq.cam, seq.par, 1000, max_keypoints=2100, precision=1)
        e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
        for t in range(4):
            e.step(*seq.frames[t])
        out[k] = e.get_state()

    ref = [None]
    run(ref, 0)
    for _ in range(2):
        res = [None, None, None]
        th = [threading.Thread(target=run, args=(res, k)) for k in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in range(3):
            np.testing.assert_array_equal(res[k][0], ref[0][0])
            np.testing.assert_array_equal(res[k][2], ref[0][2])


def test_fp32_sweep_path_is_refused_above_2048_rows(eng_mod):
    """EKF_UPDATE_PATH_SWEEP in the fp32 configuration above 2048 measurement rows changes the filter's decisions (fp32 forward
    substitution over more than 64 panels, measured at N = 5000): the engine refuses it with EKF_ERR_INVALID_ARG instead of
    returning a different filter; the same update goes through on EKF_UPDATE_PATH_AUTO."""
    from openekfmonoslam_amd.ekftypes import MATCH_DTYPE
    from openekfmonoslam_amd.engine import EkfError

    N = 1100
    seq = SyntheticSequence(N, 1)
    e = eng_mod.EkfEngine(seq.cam, seq.par, N, max_keypoints=64, precision=1)
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    e.predict()
    preds, _, _ = e.predict_measurements()
    assert 2 * len(preds) > 2048
    mt = np.zeros(len(preds), dtype=MATCH_DTYPE)
    mt["featureIndex"] = preds["featureIndex"]
    mt["keypointIndex"] = -1
    mt["imagePos"] = preds["imagePos"] + 0.25
    e.set_update_path(1)
    with pytest.raises(EkfError) as ei:
        e.update(mt)
    assert ei.value.code == 1, ei.value  # EKF_ERR_INVALID_ARG
    e.set_update_path(0)
    e.update(mt)
    x, fp, P = e.get_state()
    assert np.isfinite(P).all() and np.isfinite(fp).all()
    e.close()
