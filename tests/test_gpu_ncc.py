"""GPU parity of matcher mode B (image in, NCC over a pyramid): pyramid bytes, templates (through their effect on
the matches), the match list and the full image step, HIP engine through the C ABI vs the CPU oracle.  Integer and
index outputs (pyramid, matched pixels, match order, counts) must be identical; the fp64 state after image steps
within F64_TOL."""
import numpy as np
import pytest

from openekfmonoslam_amd.synth import SyntheticSequence
from tests.oracle_lib import ALGORITHMIC
from tests.test_gpu_parity import F32_TOL, F64_TOL, assert_state_close, eng_mod, make_pair  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels,w,h", [(1, 640, 480), (3, 322, 242), (4, 321, 243)])
def test_pyramid_identical(eng_mod, oracle_lib, channels, w, h):
    seq = SyntheticSequence(16, 1, width=w, height=h)
    img = seq.render_image(0, channels=channels)
    e, o = make_pair(eng_mod, oracle_lib, seq)
    e.upload_image(img)
    o.set_image(img)
    for lvl in range(3):
        np.testing.assert_array_equal(e.image_level(lvl), o.image_level(lvl))


def _with_templates(eng_mod, oracle_lib, seq, precision=0):
    e, o = make_pair(eng_mod, oracle_lib, seq, precision)
    img0 = seq.render_image(0)
    uv0 = seq.pixel_positions(0).astype(np.float64)
    idx = np.arange(seq.n_features)
    e.upload_image(img0)
    e.capture_templates(idx, uv0)
    o.set_image(img0)
    o.capture_templates(idx, uv0)
    return e, o


@pytest.mark.parametrize("nfeat", [12, 50, 200])
def test_match_ncc_identical(eng_mod, oracle_lib, nfeat):
    seq = SyntheticSequence(nfeat, 3)
    e, o = _with_templates(eng_mod, oracle_lib, seq)
    for t in (1, 2):
        e.predict()
        o.predict()
        e.predict_measurements()
        preds, _, _ = o.predict_measurements()
        img = seq.render_image(t)
        e.upload_image(img)
        o.set_image(img)
        mg = e.match_ncc()
        mo = o.match_ncc(preds)
        assert len(mg) == len(mo) and len(mo) > 0.6 * nfeat
        np.testing.assert_array_equal(mg["featureIndex"], mo["featureIndex"])
        np.testing.assert_array_equal(mg["keypointIndex"], mo["keypointIndex"])
        np.testing.assert_array_equal(mg["imagePos"], mo["imagePos"])
        np.testing.assert_array_equal(mg["distance"], mo["distance"])


def test_match_ncc_edge_images(eng_mod, oracle_lib):
    """flat frame (zero variance), pure noise, and predictions whose windows hang over the border"""
    seq = SyntheticSequence(40, 2, margin=2.0)
    e, o = _with_templates(eng_mod, oracle_lib, seq)
    e.predict()
    o.predict()
    e.predict_measurements()
    preds, _, _ = o.predict_measurements()
    rng = np.random.default_rng(11)
    for img in (np.full((480, 640), 200, dtype=np.uint8), rng.integers(0, 256, (480, 640), dtype=np.uint8),
                seq.render_image(1)):
        e.upload_image(img)
        o.set_image(img)
        mg, mo = e.match_ncc(), o.match_ncc(preds)
        assert len(mg) == len(mo)
        np.testing.assert_array_equal(mg["featureIndex"], mo["featureIndex"])
        np.testing.assert_array_equal(mg["imagePos"], mo["imagePos"])


@pytest.mark.parametrize("nfeat,frames", [(50, 4), (200, 2)])
def test_step_image_parity_f64(eng_mod, oracle_lib, nfeat, frames):
    seq = SyntheticSequence(nfeat, frames)
    e, o = _with_templates(eng_mod, oracle_lib, seq)
    for t in range(1, frames + 1):
        img = seq.render_image(t, channels=3 if t % 2 else 1)
        gi = e.step_image(img)
        oi = o.step_image(img, ALGORITHMIC)
        for f in ("n_predicted", "n_matches", "n_hypotheses", "n_inliers", "n_outliers", "n_rescued", "status"):
            assert getattr(gi, f) == getattr(oi, f), (t, f, getattr(gi, f), getattr(oi, f))
        assert_state_close(e, o, F64_TOL, f"image step {t}")
    _, tp, tm = e.get_map_features()
    _, tpo, tmo = o.map_features()
    np.testing.assert_array_equal(tp, tpo)
    np.testing.assert_array_equal(tm, tmo)


def test_staged_images_equal_direct_steps(eng_mod, oracle_lib):
    seq = SyntheticSequence(50, 3)
    e1, _ = _with_templates(eng_mod, oracle_lib, seq)
    e2, _ = _with_templates(eng_mod, oracle_lib, seq)
    for e in (e1, e2):
        e.set_sweep_mode(4)  # bitwise run-to-run comparisons need the launch-per-panel sweep (the persistent one batches by arrival)
    imgs = [seq.render_image(t) for t in range(1, 4)]
    e2.upload_images(imgs)
    for t in range(3):
        a = e1.step_image(imgs[t])
        b = e2.step_staged_image(t)
        assert (a.n_matches, a.n_inliers, a.n_rescued) == (b.n_matches, b.n_inliers, b.n_rescued)
    x1, f1, P1 = e1.get_state()
    x2, f2, P2 = e2.get_state()
    np.testing.assert_array_equal(x1, x2)
    np.testing.assert_array_equal(P1, P2)


def test_templates_follow_map_compaction(eng_mod, oracle_lib):
    seq = SyntheticSequence(30, 2)
    e, o = _with_templates(eng_mod, oracle_lib, seq)
    drop = np.array([3, 4, 17], dtype=np.int32)
    e.remove_features(drop)
    o.remove_features(drop)
    # the oracle keeps templates per feature slot: re-capture the survivors from frame 0 at their new indices
    keep = np.setdiff1d(np.arange(30), drop)
    o.set_image(seq.render_image(0))
    o.capture_templates(np.arange(len(keep)), seq.pixel_positions(0)[keep].astype(np.float64))
    e.predict()
    o.predict()
    e.predict_measurements()
    preds, _, _ = o.predict_measurements()
    img = seq.render_image(1)
    e.upload_image(img)
    o.set_image(img)
    mg, mo = e.match_ncc(), o.match_ncc(preds)
    np.testing.assert_array_equal(mg["featureIndex"], mo["featureIndex"])
    np.testing.assert_array_equal(mg["imagePos"], mo["imagePos"])


def test_ncc_requires_an_image(eng_mod, oracle_lib):
    seq = SyntheticSequence(12, 1)
    e, _ = make_pair(eng_mod, oracle_lib, seq)
    e.predict()
    e.predict_measurements()
    with pytest.raises(eng_mod.EkfError):
        e.match_ncc()


@pytest.mark.parametrize("nfeat,max_new", [(40, 25), (40, 500), (150, 60)])
def test_detect_new_features_identical(eng_mod, oracle_lib, nfeat, max_new):
    """masked corner candidates + zone heuristic: device/host engine path == oracle (integer measure: identical)"""
    seq = SyntheticSequence(nfeat, 2)
    e, o = _with_templates(eng_mod, oracle_lib, seq)
    # no predictions yet (EKF::init: detectNewImageFeatures(image, noPredictions, ...), EKF.cpp:196)
    none = np.zeros(0, dtype=oracle_lib.PREDICTION_DTYPE)
    np.testing.assert_array_equal(e.detect_new_features(max_new), o.detect_new_features(none, max_new))
    # with the predictions of a frame masking their ellipses
    img = seq.render_image(1)
    e.upload_image(img)
    o.set_image(img)
    e.predict()
    o.predict()
    e.predict_measurements()
    preds, _, _ = o.predict_measurements()
    ug, uo = e.detect_new_features(max_new), o.detect_new_features(preds, max_new)
    assert len(uo) > 0
    np.testing.assert_array_equal(ug, uo)
    # a high threshold leaves nothing
    assert len(e.detect_new_features(max_new, min_response=1e30)) == 0


def test_image_pipeline_init_track_and_grow(eng_mod, oracle_lib):
    """EKF::init(image) / step(image) / map management in matcher mode B, engine vs oracle, same calls on both:
    reset -> detect -> add -> capture templates; frames: step_image; then remove bad, convert, detect, add."""
    seq = SyntheticSequence(60, 4)
    e = eng_mod.EkfEngine(seq.cam, seq.par, 128)
    o = oracle_lib.Oracle(seq.cam, seq.par, 128)
    img0 = seq.render_image(0)
    e.reset()
    o.reset()
    e.upload_image(img0)
    o.set_image(img0)
    none = np.zeros(0, dtype=oracle_lib.PREDICTION_DTYPE)
    uv = e.detect_new_features(40)
    np.testing.assert_array_equal(uv, o.detect_new_features(none, 40))
    assert len(uv) == 40
    e.add_features(uv)
    for p in uv:
        o.add_feature(p)
    idx = np.arange(len(uv))
    e.capture_templates(idx, uv)
    o.capture_templates(idx, uv)
    # the synthetic camera moves with a known velocity; give both filters the same prior
    x, fp, P = e.get_state()
    x[7:10] = [0.01, 0.0, 0.002]
    x[10:13] = [2.22e-16, 0.002, 2.22e-16]
    ftype = np.full(len(uv), 2, dtype=np.int32)
    e.set_state(x, fp, ftype, None, P)
    o.set_state(x, fp, ftype, None, P)
    e.upload_image(img0)
    e.capture_templates(idx, uv)
    for t in range(1, 4):
        img = seq.render_image(t, outlier_fraction=0.0)
        gi = e.step_image(img)
        oi = o.step_image(img, ALGORITHMIC)
        for f in ("n_predicted", "n_matches", "n_inliers", "n_rescued", "status"):
            assert getattr(gi, f) == getattr(oi, f), (t, f, getattr(gi, f), getattr(oi, f))
        assert gi.n_matches >= 25
    assert_state_close(e, o, 1e-8, "image pipeline, 3 frames")
    # map management, reference order (EKF.cpp:575-612)
    assert e.remove_bad_features() == o.remove_bad_features()
    assert e.convert_inverse_depth_to_depth() == o.convert_inverse_depth_to_depth()
    preds = o.predict_measurements()[0]  # the oracle is handed this step's gates explicitly
    e.predict_measurements()
    want = 10
    uvn = e.detect_new_features(want)
    np.testing.assert_array_equal(uvn, o.detect_new_features(preds, want))
    n0 = e.N
    e.add_features(uvn)
    for p in uvn:
        o.add_feature(p)
    e.capture_templates(np.arange(n0, n0 + len(uvn)), uvn)
    assert e.N == o.N == n0 + len(uvn)
    assert_state_close(e, o, 1e-8, "after map management")


def test_real_frames_engine_equals_oracle(eng_mod, oracle_lib):
    """tests/golden/s3_frames: eight frames of the reference's sample sequence (reduced to 320x240 gray by
    tests/golden/make_s3_frames.py).  Same calls on engine and oracle: init on frame 0 (detect, add, templates), then
    image steps with the device map management in the reference's order.  Picks, matches and counts identical; state and
    covariance within 1e-8 after seven frames of real imagery."""
    import os

    from PIL import Image

    from openekfmonoslam_amd.ekftypes import s3_camera, s3_params

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s3_frames")
    frames = [np.asarray(Image.open(os.path.join(d, f"{k:05d}.png"))) for k in range(8)]
    assert frames[0].shape == (240, 320)
    cam, par = s3_camera(320, 240), s3_params()
    e = eng_mod.EkfEngine(cam, par, 96)
    o = oracle_lib.Oracle(cam, par, 96)
    e.reset()
    o.reset()
    e.upload_image(frames[0])
    o.set_image(frames[0])
    none = np.zeros(0, dtype=oracle_lib.PREDICTION_DTYPE)
    uv = e.detect_new_features(40, min_response=1e10)
    np.testing.assert_array_equal(uv, o.detect_new_features(none, 40, min_response=1e10))
    assert len(uv) == 40
    e.add_features(uv)
    for p in uv:
        o.add_feature(p)
    e.capture_templates(np.arange(40), uv)
    o.capture_templates(np.arange(40), uv)
    for t in range(1, 8):
        gi = e.step_image(frames[t])
        oi = o.step_image(frames[t], ALGORITHMIC)
        for f in ("n_predicted", "n_matches", "n_hypotheses", "n_inliers", "n_outliers", "n_rescued", "status"):
            assert getattr(gi, f) == getattr(oi, f), (t, f, getattr(gi, f), getattr(oi, f))
        assert gi.n_matches >= 30
    assert_state_close(e, o, 1e-8, "seven real frames")
    x, _, _ = e.get_state(want_P=False)
    assert x[0] < -0.005  # the camera of this sequence slides sideways
