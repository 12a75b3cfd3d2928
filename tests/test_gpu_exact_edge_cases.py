"""Edge cases of the path in the EKF_PRECISION_F32_EXACT configuration (fp32 storage of P, fp64 B from int8 digit planes, exact
int8 downdate -- the configuration bench.py headlines), HIP engine through the C ABI vs the fp64 oracle: update sizes that
straddle every blocking boundary (32-row panels, 16-row digit groups, the 128-wide tiles of the downdate) on both ways of
forming B, empty and ragged inputs, a frame of outliers, mixed XYZ / inverse-depth maps (columns of three and of six per
feature), map management between frames, an indefinite innovation covariance, a long run, and two engines on two streams.
One tolerance: every block AND every feature parameter within 1e-5 (parity_metric.over_tolerance)."""
import threading

import numpy as np
import pytest

from openekfmonoslam_amd.ekftypes import KEYPOINT_DTYPE, MATCH_DTYPE
from openekfmonoslam_amd.synth import SyntheticSequence
from parity_metric import F32_TOL, over_tolerance, parity_report
from tests.oracle_lib import ALGORITHMIC, align_to_matches
from tests.test_gpu_edge_cases import INFO, _matches_from_predictions, _same_info  # noqa: F401
from tests.test_gpu_parity import eng_mod, make_pair  # noqa: F401

pytestmark = pytest.mark.gpu
EXACT = 2


def assert_parity(e, o, what, n_features=0):
    x, fp, P = e.get_state()
    be = parity_report(x, fp, P, o.x13(), o.feature_pos(), o.P())
    bad = over_tolerance(be, F32_TOL, n_features, componentwise=True)
    assert not bad, (what, bad)
    # the exact downdate leaves P bitwise symmetric
    assert np.array_equal(P, P.T), what
    return be


@pytest.mark.parametrize("path", [pytest.param(1, id="sweep"), pytest.param(2, id="gemm")])
@pytest.mark.parametrize("M", [1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 160])
def test_update_sizes_across_block_boundaries_exact(eng_mod, oracle_lib, M, path):
    """m = 2M rows of S, 2 .. 320: below / at / above a 16-row digit group, a 32-row panel, a 128-column tile; rows of B
    inside the sweep (from digit planes) or by inverse + GEMM"""
    seq = SyntheticSequence(170, 1, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=EXACT)
    e.set_update_path(path)
    e.predict()
    o.predict()
    e.predict_measurements()
    preds, Hs, Hf = o.predict_measurements()
    mo = _matches_from_predictions(preds, M)
    mp, mHs, mHf = align_to_matches(preds, Hs, Hf, mo)
    assert o.update(mo, mp, mHs, mHf, ALGORITHMIC) == 0
    e.update(mo)
    assert_parity(e, o, f"update with M = {M}", 170)


@pytest.mark.parametrize("M", [1, 16, 17, 32, 33, 48, 49, 64, 65, 80, 96, 97, 128, 129, 160])
def test_update_sizes_two_panels_per_launch_exact(eng_mod, oracle_lib, M):
    """the sweep in its two-panels-per-launch form with the rows of B from digit planes (chol_bplanes.h b_pair_rows_planes):
    odd and even panel counts, a short last panel in either half of a pair"""
    seq = SyntheticSequence(170, 1, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=EXACT)
    e.set_update_path(1)
    e.set_sweep_mode(0)
    e.predict()
    o.predict()
    e.predict_measurements()
    preds, Hs, Hf = o.predict_measurements()
    mo = _matches_from_predictions(preds, M)
    mp, mHs, mHf = align_to_matches(preds, Hs, Hf, mo)
    assert o.update(mo, mp, mHs, mHf, ALGORITHMIC) == 0
    e.update(mo)
    assert_parity(e, o, f"update with M = {M}, two panels per launch", 170)


@pytest.mark.parametrize("mode", [pytest.param(0, id="pairs"), pytest.param(1, id="single")])
def test_sweep_modes_exact(eng_mod, oracle_lib, mode):
    """ekf_set_sweep_mode in the exact configuration (the rows of B from digit planes exist for one panel per launch: a
    request for pairs must still give the right answer)"""
    seq = SyntheticSequence(170, 2, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=EXACT)
    e.set_sweep_mode(mode)
    for t in range(2):
        gi = e.step(*seq.frames[t])
        oi = o.step(*seq.frames[t], ALGORITHMIC)
        _same_info(gi, oi, f"frame {t}")
    assert_parity(e, o, f"sweep mode {mode}", 170)


@pytest.mark.parametrize("mode,persistent", [pytest.param(2, True, id="auto"), pytest.param(3, True, id="persistent"),
                                             pytest.param(4, False, id="launches")])
def test_sweep_mode_decides_the_launch_count_exact(eng_mod, oracle_lib, mode, persistent):
    """EKF_SWEEP_AUTO takes the persistent sweep on a map of fewer than 8192 state columns (ONE launch per update:
    ekf_timing_sweep_launches == updates), EKF_SWEEP_LAUNCHES the launch-per-panel sweep (a launch per panel or pair); both give the
    oracle's answer."""
    seq = SyntheticSequence(170, 2, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=EXACT)
    e.set_sweep_mode(mode)
    e.timing(True)
    e.timing_reset()
    for t in range(2):
        gi = e.step(*seq.frames[t])
        oi = o.step(*seq.frames[t], ALGORITHMIC)
        _same_info(gi, oi, f"frame {t}")
    sw = e.sweep_timing()
    assert sw["updates"] > 0 and sw["panels"] > sw["updates"]
    assert (sw["launches"] == sw["updates"]) == persistent, sw
    assert_parity(e, o, f"sweep mode {mode}", 170)


def test_auto_sweep_keeps_the_launch_per_panel_sweep_on_wide_maps(eng_mod):
    """From 8192 state columns on (N >= 1364) EKF_SWEEP_AUTO leaves the persistent sweep (its B workers would take several blocks of
    columns each: N = 2000 measured 20.4 against 14.0 us per panel); EKF_SWEEP_PERSISTENT still forces it and gives the same filter
    to rounding (one update of 1200 rows through the stage functions)."""
    seq = SyntheticSequence(1400, 1, width=1280, height=720)
    out = []
    for mode in (2, 3):
        e = eng_mod.EkfEngine(seq.cam, seq.par, 1400, max_keypoints=len(seq.frames[0][0]) + 64, precision=EXACT)
        e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
        e.set_sweep_mode(mode)
        e.timing(True)
        e.timing_reset()
        e.predict()
        preds, _, _ = e.predict_measurements()
        e.update(_matches_from_predictions(preds, 600))  # 1200 rows: the rows of B inside the sweep
        sw = e.sweep_timing()
        x, fp, P = e.get_state()
        out.append((sw, x, fp, P))
        e.close()
    (swa, xa, fa, Pa), (swp, xp, fpp, Pp) = out
    assert swa["launches"] > swa["updates"] and swp["launches"] == swp["updates"] == 1, (swa, swp)
    # (two orders of the same fp64 sums under one fp32 rounding of P per entry: far inside the 1e-5 either holds against the oracle)
    assert np.abs(xa - xp).max() <= 1e-7 and np.abs(fa - fpp).max() <= 1e-7, (np.abs(xa - xp).max(), np.abs(fa - fpp).max())
    assert np.abs(Pa - Pp).max() <= 1e-6 * np.abs(Pa).max()


def test_ragged_inputs_exact(eng_mod, oracle_lib, seq12):
    """no keypoints, then a normal frame; camera turned away; a frame in which every keypoint is displaced"""
    e, o = make_pair(eng_mod, oracle_lib, seq12, precision=EXACT)
    kps = np.zeros(0, dtype=KEYPOINT_DTYPE)
    desc = np.zeros((0, 32), dtype=np.uint8)
    _same_info(e.step(kps, desc), o.step(kps, desc, ALGORITHMIC), "empty frame")
    assert_parity(e, o, "empty frame (prediction only)", 12)
    _same_info(e.step(*seq12.frames[1]), o.step(*seq12.frames[1], ALGORITHMIC), "frame after the empty one")
    assert_parity(e, o, "frame after the empty one", 12)
    # every keypoint displaced: one-point hypothesis, re-prediction / rescue path
    e, o = make_pair(eng_mod, oracle_lib, seq12, precision=EXACT)
    kps, desc = seq12.frames[0]
    bad = kps.copy()
    rng = np.random.default_rng(9)
    bad["x"] += rng.uniform(-6, 6, len(bad)).astype(np.float32)
    bad["y"] += rng.uniform(-6, 6, len(bad)).astype(np.float32)
    _same_info(e.step(bad, desc), o.step(bad, desc, ALGORITHMIC), "scattered keypoints")
    assert_parity(e, o, "scattered keypoints", 12)
    # nothing visible
    e, o = make_pair(eng_mod, oracle_lib, seq12, precision=EXACT)
    x = np.array(seq12.x13)
    x[3:7] = [0.0, 0.0, 1.0, 0.0]
    for f in (e, o):
        f.set_state(x, seq12.feature_pos, seq12.feature_type, seq12.feature_desc, seq12.P0)
    gi = e.step(*seq12.frames[0])
    _same_info(gi, o.step(*seq12.frames[0], ALGORITHMIC), "nothing visible")
    assert gi.n_predicted == 0
    assert_parity(e, o, "nothing visible", 12)


def test_empty_map_exact(eng_mod, oracle_lib, seq12):
    """N = 0: the 13-state filter alone, then features added by the engine's own map management and two more frames"""
    e = eng_mod.EkfEngine(seq12.cam, seq12.par, 16, precision=EXACT)
    o = oracle_lib.Oracle(seq12.cam, seq12.par, 16)
    e.reset()
    o.reset()
    for t in range(2):
        _same_info(e.step(*seq12.frames[t]), o.step(*seq12.frames[t], ALGORITHMIC), f"empty map, frame {t}")
    assert e.n == 13
    assert_parity(e, o, "empty map")


def test_mixed_depth_and_inverse_depth_map_exact(eng_mod, oracle_lib):
    """columns of three (XYZ) and of six (inverse depth) per feature in one map: one conversion per frame after the updates
    (EKF.cpp:594, threshold raised so that every call converts the first remaining inverse-depth feature), then the next
    step; the digit planes, their a-priori column scales and the tiles of the downdate must not care where a feature's
    columns start.  Three frames: converting features that are NOT linear yet is an ill-conditioned stress (the fp64 engine
    itself agrees with the oracle to 1e-8 instead of 1e-12 on it, tests/test_gpu_map_management.py) that multiplies any
    rounding of P by 3-10 per frame -- the fp32-storage floor (scripts/dbg_mixed_map.py) is 1e-8, 4e-8, 3e-6 on these three
    frames, the exact configuration 2e-8, 1e-7, 6e-6, and both leave 1e-5 two frames later."""
    seq = SyntheticSequence(50, 3)
    seq.par.inverseDepthLinearityIndexThreshold = 1e9
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=EXACT)
    for t, (kps, desc) in enumerate(seq.frames):
        _same_info(e.step(kps, desc), o.step(kps, desc, ALGORITHMIC), f"frame {t}")
        assert e.convert_inverse_depth_to_depth() == o.convert_inverse_depth_to_depth() == t
        be = assert_parity(e, o, f"mixed map, frame {t}", 50)
        print(f"mixed map frame {t}:", {k: f"{v:.1e}" for k, v in be.items()})
    assert e.n == 313 - 3 * 3
    # removeBadMapFeatures (MapManagement.cpp:279): same features go on both sides (the removal takes the largest entries of P
    # with it, so the norm-wise measure of P is not comparable across it: the state blocks are)
    assert e.remove_bad_features() == o.remove_bad_features()
    x, fp, P = e.get_state()
    assert P.shape == o.P().shape and np.array_equal(P, P.T)
    bad = over_tolerance(parity_report(x, fp, P, o.x13(), o.feature_pos(), o.P()), F32_TOL, 50)
    assert not {k: v for k, v in bad.items() if not k.startswith("P_")}, bad


def test_remove_features_between_frames_exact(eng_mod, oracle_lib, seq50):
    """features removed from the middle of the map between frames (MapManagement.cpp:212): P is compacted in fp32, the next
    update runs on the shorter state"""
    e, o = make_pair(eng_mod, oracle_lib, seq50, precision=EXACT)
    _same_info(e.step(*seq50.frames[0]), o.step(*seq50.frames[0], ALGORITHMIC), "frame 0")
    gone = [3, 4, 17, 30, 49]
    e.remove_features(gone)
    o.remove_features(gone)
    assert e.n == 13 + 6 * 45
    for t in (1, 2):
        kps, desc = seq50.frames[t]
        _same_info(e.step(kps, desc), o.step(kps, desc, ALGORITHMIC), f"frame {t} after removal")
    assert_parity(e, o, "after removal", 45)


def test_indefinite_innovation_covariance_is_reported_exact(eng_mod, seq12):
    e = eng_mod.EkfEngine(seq12.cam, seq12.par, 16, max_keypoints=128, precision=EXACT)
    P = -1e3 * np.eye(seq12.state_dim)
    e.set_state(seq12.x13, seq12.feature_pos, seq12.feature_type, seq12.feature_desc, P)
    e.predict()
    preds, _, _ = e.predict_measurements()
    m = np.zeros(4, dtype=MATCH_DTYPE)
    m["featureIndex"] = preds["featureIndex"][:4]
    m["imagePos"] = preds["imagePos"][:4] + 0.5
    x0, fp0, P0 = e.get_state()
    with pytest.raises(eng_mod.EkfError) as ei:
        e.update(m)
    assert ei.value.code == 3
    # ... and the update is SKIPPED, as the reference skips it (cv::invert returns zeros: K = 0, Update.cpp:101-108): the kernels
    # behind a failed factorisation leave x and P exactly as they were
    x1, fp1, P1 = e.get_state()
    assert np.array_equal(x0, x1) and np.array_equal(fp0, fp1) and np.array_equal(P0, P1)


@pytest.mark.parametrize("mode", [pytest.param(2, id="persistent"), pytest.param(4, id="launches")])
def test_rows_of_B_that_do_not_fit_their_column_scale_are_reported_exact(eng_mod, seq50, mode):
    """Round-5 review item 6: the rows of B are cut into digits under an A-PRIORI column scale, |B_kj| <= sqrt(P_jj) -- true for a
    positive semi-definite P.  A covariance uploaded with |P_ij| > sqrt(P_ii P_jj) breaks the bound; digits that wrap would downdate P
    with garbage in silence.  The cut now raises EKF_ERR_NON_FINITE (4), the kernels behind the sweep leave the filter untouched, and
    the update reports it.  One inverse depth with variance 1e-20 but covariances of 1e-6 with the camera position: S stays
    positive definite, the rows of B of that column are ~1e6 x their scale."""
    seq = seq50
    e = eng_mod.EkfEngine(seq.cam, seq.par, 64, max_keypoints=512, precision=EXACT)
    e.set_sweep_mode(mode)
    P = np.array(seq.P0, copy=True)
    j = 13 + 6 * 3 + 5
    P[j, j] = 1e-20
    P[j, :3] = P[:3, j] = 1e-6
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P)
    e.predict()
    preds, _, _ = e.predict_measurements()
    m = _matches_from_predictions(preds, 40)
    x0, fp0, P0 = e.get_state()
    with pytest.raises(eng_mod.EkfError) as ei:
        e.update(m)
    assert ei.value.code == 4, ei.value
    x1, fp1, P1 = e.get_state()
    assert np.array_equal(x0, x1) and np.array_equal(fp0, fp1) and np.array_equal(P0, P1)
    # the engine is usable afterwards: a proper covariance, the same update
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    e.predict()
    e.predict_measurements()
    e.update(m)


@pytest.mark.parametrize("path", [pytest.param(1, id="sweep"), pytest.param(2, id="gemm")])
def test_n200_exact_frames_vs_oracle(eng_mod, oracle_lib, path):
    """BASELINE configs[1] map size in the exact configuration, three frames, both ways of forming B"""
    seq = SyntheticSequence(200, 3)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=EXACT)
    e.set_update_path(path)
    for t in range(3):
        _same_info(e.step(*seq.frames[t]), o.step(*seq.frames[t], ALGORITHMIC), f"frame {t}")
    be = assert_parity(e, o, "N = 200, three frames", 200)
    print(f"N=200 exact, update path {path}:", {k: f"{v:.2e}" for k, v in be.items()})


def test_exact_drift_over_90_frames(eng_mod, oracle_lib):
    """SURVEY 8(d) drift check in the exact configuration: N = 100, 90 frames against the fp64 oracle, identical decisions"""
    seq = SyntheticSequence(100, 90)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=EXACT)
    for t in range(90):
        _same_info(e.step(*seq.frames[t]), o.step(*seq.frames[t], ALGORITHMIC), f"frame {t}")
    be = assert_parity(e, o, "90 frames", 100)
    print("exact configuration after 90 frames:", {k: f"{v:.2e}" for k, v in be.items()})


@pytest.mark.parametrize("mode", [pytest.param(4, id="launches_bitwise"), pytest.param(2, id="persistent")])
def test_concurrent_exact_engines_give_the_sequential_result(eng_mod, mode):
    """two exact engines stepped from two host threads (their own streams, work lists and digit-plane buffers) give the result of
    running them one after the other.  With the launch-per-panel sweep: bit for bit.  With the persistent sweep (the default): to
    1e-12 -- its tile tasks are picked up by arrival, and which role applies a panel to a tile first decides the last bit of an fp64
    sum of S; a sweep that timed out under the other engine's kernels is run again on the launch-per-panel path, whose sums are
    grouped differently again.  scripts/concurrency_probe.py: 2 of 3000 concurrent runs differ from the run alone, in the last bits of
    x (1e-16 relative), none of 3000 with the launch-per-panel sweep."""
    seq = SyntheticSequence(300, 4)

    def run(out, k):
        e = eng_mod.EkfEngine(seq.cam, seq.par, 300, max_keypoints=len(seq.frames[0][0]) + 64, precision=EXACT)
        e.set_sweep_mode(mode)
        e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
        for t in range(4):
            e.step(*seq.frames[t])
        out[k] = e.get_state()
        e.close()

    ref = {}
    run(ref, 0)
    got = {}
    th = [threading.Thread(target=run, args=(got, k)) for k in (1, 2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in (1, 2):
        for a, b in zip(ref[0], got[k]):
            a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
            if mode == 4:
                assert np.array_equal(a, b), (k, float(np.max(np.abs(a - b))))
            else:
                assert np.max(np.abs(a - b)) <= 1e-12 * max(np.max(np.abs(a)), 1e-300), (k, float(np.max(np.abs(a - b))))


@pytest.mark.parametrize("nfeat,frames,precision", [pytest.param(1000, 3, EXACT, id="n1000_f32_stored"), pytest.param(1400, 2, EXACT, id="n1400_above_2048_rows"),
                                                    pytest.param(1400, 2, 3, id="n1400_f64_stored")])
def test_skipping_the_zero_pieces_of_plane_0_changes_no_bit(eng_mod, nfeat, frames, precision):
    """the downdate (and, above 2048 rows, the int8 GEMM B = inv(L) G) leave out the digit products whose plane-0 operand piece is
    all zeros; with the hook that makes them multiply everything (include/ekf_test_hooks.h) the state, every feature and P must be
    the same bit for bit.  Fresh maps: most pieces ARE zero, the skipping step is the one that runs; N = 1400: updates of ~2500
    rows, whose zero-piece masks span several 64-step chunks.  (Launch-per-panel sweep: the persistent one orders its sums by arrival.)"""
    seq = SyntheticSequence(nfeat, frames)
    out = []
    for dense in (False, True):
        e = eng_mod.EkfEngine(seq.cam, seq.par, nfeat + 8, max_keypoints=len(seq.frames[0][0]) + 64, precision=precision)
        e.set_sweep_mode(4)
        e.dense_products(dense)
        e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
        infos = [e.step(*seq.frames[t]) for t in range(frames)]
        nz, tot = e.plane0_pieces()
        out.append((e.get_state(), [(int(i.n_inliers), int(i.n_rescued)) for i in infos], nz, tot))
        e.close()
    (sa, ia, nz, tot), (sb, ib, _, _) = out
    assert ia == ib
    if nfeat > 1024:
        assert 2 * max(ia[0]) > 2048, ia  # an update above 2048 rows was among them
    assert tot > 0 and 2 * nz < tot, (nz, tot)  # the data did have zero pieces to skip
    for a, b in zip(sa, sb):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_exact_configuration_refuses_maps_whose_int32_level_sums_could_wrap(eng_mod):
    """kernels_pexact.hip: the int32 level sums of the exact downdate are exact while 5 * 2^14 * m < 2^31, i.e. up to 26208 rows of
    B (round-4 ADVICE): ekf_engine_create returns EKF_ERR_INVALID_ARG for EKF_PRECISION_F32_EXACT with 2 * max_features above that,
    before anything is allocated; the largest admissible capacity is accepted by the argument check (13104 features)."""
    seq = SyntheticSequence(8, 1)
    with pytest.raises(eng_mod.EkfError) as ei:
        eng_mod.EkfEngine(seq.cam, seq.par, 13105, max_keypoints=64, precision=EXACT)
    assert ei.value.code == 1  # EKF_ERR_INVALID_ARG


@pytest.mark.parametrize("precision", [EXACT, 3, 0])
def test_stalled_persistent_sweep_is_retried_on_the_launch_per_panel_sweep(eng_mod, oracle_lib, precision):
    """Watchdog + retry of the persistent Cholesky sweep (csrc/chol_persist.h; round-5 review item 6): ekf_debug_stall_next_sweep
    (include/ekf_test_hooks.h) makes the next sweep run WITHOUT its chain workgroup, so every tile worker and every B worker waits
    for an inverse that is never published.  The bounded waits end the kernel (30 ms; asserted here as < 5 s of wall time) with the
    sticky code EKF_ERR_TIMEOUT, the kernels behind the sweep leave x and P untouched, and ekf_update runs the SAME update again
    from the same P on the launch-per-panel sweep: the caller gets EKF_OK, the oracle's result, and one retry on the counter."""
    import time
    seq = SyntheticSequence(170, 1, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=precision)
    e.predict()
    e.predict_measurements()
    o.predict()
    preds, Hs, Hf = o.predict_measurements()
    mo = _matches_from_predictions(preds, 129)
    mp, mHs, mHf = align_to_matches(preds, Hs, Hf, mo)
    assert o.update(mo, mp, mHs, mHf, ALGORITHMIC) == 0
    assert e.sweep_retries == 0
    assert e.L.ekf_debug_stall_next_sweep(e.h) == 0
    t0 = time.time()
    e.update(mo)  # no error surfaces
    e.synchronize()
    assert time.time() - t0 < 5.0
    assert e.sweep_retries == 1
    x, fp, P = e.get_state()
    be = parity_report(x, fp, P, o.x13(), o.feature_pos(), o.P())
    assert not over_tolerance(be, 1e-9 if precision == 0 else F32_TOL, 170, componentwise=True), be
    if precision != 0:
        assert np.array_equal(P, P.T)  # (the first downdate after an upload symmetrises: the retry must do it, the frozen one must not have)


@pytest.mark.parametrize("async_errors", [False, True], ids=["sync", "async"])
@pytest.mark.parametrize("which", [0, 1, 2, 3], ids=["li_frame0", "hi_frame0", "li_frame1", "hi_frame1"])
def test_stalled_sweep_inside_a_step_is_invisible_to_the_caller(eng_mod, oracle_lib, which, async_errors):
    """The same stall inside EKF::step (the staged-frames path bench.py times), on the first or the second update of a frame, with
    and without ekf_set_async_errors: the failed update is seen at the step's next read-back (or, async, at the NEXT step's first
    one -- that step's prediction must not have touched the filter), everything enqueued behind it was frozen, the update is run
    again and the stages behind it repeated.  Decisions, state, covariance and the map's timesMatched equal the oracle's after
    every frame; EkfStepInfo.n_sweep_retries reports the retry in the step that performed it."""
    seq = SyntheticSequence(170, 3, seed=5)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=EXACT)
    e.upload_frames(seq.frames)
    e.set_async_errors(async_errors)
    e.stall_sweep(which)  # persistent sweeps from now on: frame 0 LI, frame 0 HI, frame 1 LI, ...
    retried = 0
    for t in range(3):
        gi = e.step_frame(t)
        oi = o.step(*seq.frames[t], ALGORITHMIC)
        assert gi.status == 0
        _same_info(gi, oi, f"frame {t}")
        assert oi.n_inliers > 0 and oi.n_rescued > 0, "the scene must exercise both updates"
        retried += gi.n_sweep_retries
    e.synchronize()
    assert e.sweep_retries == 1, e.sweep_retries
    if not async_errors:
        assert retried == 1
    assert_parity(e, o, f"stall {which}, async {async_errors}", 170)
    _, tp, tm = e.get_map_features()
    _, otp, otm = o.map_features()
    assert np.array_equal(tm, otm) and np.array_equal(tp, otp)


def test_persistent_sweep_falls_back_when_its_grid_does_not_fit(eng_mod, oracle_lib):
    """the persistent sweep's grid layout is decided BEFORE the gather / assembly launch (which leaves the first block's
    factorisation to the chain workgroup of a persistent sweep); an update whose layout does not fit the resident workgroups
    must take the launch-per-panel sweep, not fail.  The largest update the flags cover (m = 2048 rows) on this device, forced
    persistent, against the launch-per-panel result: identical decisions either way and no error."""
    seq = SyntheticSequence(1024, 1, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    res = []
    for mode in (3, 4):
        e = eng_mod.EkfEngine(seq.cam, seq.par, 1024, max_keypoints=64, precision=EXACT)
        e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
        e.set_sweep_mode(mode)
        e.set_update_path(1)
        e.predict()
        preds, _, _ = e.predict_measurements()
        mo = _matches_from_predictions(preds, 1024)
        e.update(mo)
        res.append(e.get_state())
        assert e.sweep_retries == 0
        e.close()
    for a, b in zip(res[0], res[1]):
        assert np.allclose(a, b, rtol=0, atol=1e-6 * np.abs(b).max())


@pytest.mark.parametrize("M", [1, 17, 33, 64, 129, 160])
def test_update_sizes_fp64_stored_exact(eng_mod, oracle_lib, M):
    """EKF_PRECISION_F64_EXACT (what EKF_PRECISION_AUTO selects above 1024 features): the same exact int8 update on an fp64-stored
    covariance -- the downdate kernel's fp64 epilogue (half a block staged at a time, both images as 16-byte stores) across the
    blocking boundaries; P stays bitwise symmetric; held to 1e-7 (38-bit digits of B under a-priori column scales; no storage rounding)"""
    seq = SyntheticSequence(170, 1, outlier_fraction=0.0, distractors_per_feature=0.0, max_bit_flips=0)
    e, o = make_pair(eng_mod, oracle_lib, seq, precision=3)
    assert e.precision == 3
    e.predict()
    o.predict()
    e.predict_measurements()
    preds, Hs, Hf = o.predict_measurements()
    mo = _matches_from_predictions(preds, M)
    mp, mHs, mHf = align_to_matches(preds, Hs, Hf, mo)
    assert o.update(mo, mp, mHs, mHf, ALGORITHMIC) == 0
    e.update(mo)
    x, fp, P = e.get_state()
    be = parity_report(x, fp, P, o.x13(), o.feature_pos(), o.P())
    assert not over_tolerance(be, 1e-7, 170, componentwise=True), be
    assert np.array_equal(P, P.T)


def test_auto_precision_resolves_by_capacity_and_steps_fp64_stored(eng_mod, oracle_lib):
    """EKF_PRECISION_AUTO: fp32 storage up to 1024 features, fp64 storage above (ekf_get_precision reports the choice); three frames
    of an fp64-stored exact engine (the seeded P0 is symmetric to rounding only: the first downdate symmetrises) against the oracle at 1e-7"""
    seq = SyntheticSequence(150, 3)
    small = eng_mod.EkfEngine(seq.cam, seq.par, 1024, max_keypoints=64, precision=4)
    assert small.precision == 2
    small.close()
    big = eng_mod.EkfEngine(seq.cam, seq.par, 1025, max_keypoints=len(seq.frames[0][0]) + 64, precision=4)
    huge = eng_mod.EkfEngine(seq.cam, seq.par, 2049, max_keypoints=64, precision=4)
    assert huge.precision == 0  # all fp64 above 2048 features
    huge.close()
    assert big.precision == 3
    o = oracle_lib.Oracle(seq.cam, seq.par, 160)
    P0 = seq.P0
    big.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
    o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
    for t in range(3):
        _same_info(big.step(*seq.frames[t]), o.step(*seq.frames[t], ALGORITHMIC), f"frame {t}")
    x, fp, P = big.get_state()
    be = parity_report(x, fp, P, o.x13(), o.feature_pos(), o.P())
    assert not over_tolerance(be, 1e-7, 150, componentwise=True), be
    assert np.array_equal(P, P.T)
    big.close()
