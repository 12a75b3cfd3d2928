"""Matcher mode B (NCC over a pyramid) -- CPU checks of the oracle's definition against independent numpy
restatements and on rendered sequences.  The reference has no such matcher; these tests pin the definition the HIP
kernels are compared with."""
import numpy as np
import pytest

from openekfmonoslam_amd.synth import SyntheticSequence
from tests.oracle_lib import ALGORITHMIC, Oracle


def _seeded(seq):
    o = Oracle(seq.cam, seq.par, seq.n_features)
    o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    return o


def _np_pyramid(gray):
    out = [gray]
    for _ in range(2):
        g = out[-1].astype(np.int32)
        h, w = g.shape[0] // 2, g.shape[1] // 2
        g = g[: 2 * h, : 2 * w]
        out.append(((g[0::2, 0::2] + g[0::2, 1::2] + g[1::2, 0::2] + g[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    return out


@pytest.mark.parametrize("channels", [1, 3, 4])
def test_pyramid_matches_numpy(oracle_lib, channels):
    seq = SyntheticSequence(20, 1, width=322, height=242)  # odd halves at level 1: 161 x 121 -> 80 x 60
    img = seq.render_image(0, channels=channels)
    o = _seeded(seq)
    o.set_image(img)
    if channels == 1:
        gray = img
    else:
        a = img.astype(np.int32)
        r, g, b = (a[..., 2], a[..., 1], a[..., 0]) if channels == 3 else (a[..., 0], a[..., 1], a[..., 2])
        gray = ((77 * r + 150 * g + 29 * b + 128) >> 8).astype(np.uint8)
    for lvl, ref in enumerate(_np_pyramid(gray)):
        got = o.image_level(lvl)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref)


def test_templates_are_clamped_windows(oracle_lib):
    seq = SyntheticSequence(12, 1)
    img = seq.render_image(0)
    o = _seeded(seq)
    o.set_image(img)
    uv = np.array([[0.2, 0.4], [639.0, 479.0], [100.6, 50.5], [321.49, 7.0]])
    o.capture_templates([0, 1, 2, 3], uv)
    t = o.templates()
    pyr = _np_pyramid(img)
    for i in range(4):
        for lvl in range(3):
            cx = int(np.floor((uv[i, 0] + 0.5) / (1 << lvl)))
            cy = int(np.floor((uv[i, 1] + 0.5) / (1 << lvl)))
            h, w = pyr[lvl].shape
            ys = np.clip(np.arange(cy - 5, cy + 6), 0, h - 1)
            xs = np.clip(np.arange(cx - 5, cx + 6), 0, w - 1)
            assert np.array_equal(t[i, lvl], pyr[lvl][np.ix_(ys, xs)])


def test_ncc_finds_the_pasted_patches(oracle_lib):
    seq = SyntheticSequence(50, 3)
    o = _seeded(seq)
    o.set_image(seq.render_image(0))
    o.capture_templates(np.arange(50), seq.pixel_positions(0).astype(np.float64))
    o.predict()
    preds, _, _ = o.predict_measurements()
    o.set_image(seq.render_image(1, outlier_fraction=0.0))
    m = o.match_ncc(preds)
    assert len(m) >= 40
    truth = seq.pixel_positions(1)
    err = np.abs(m["imagePos"] - truth[m["featureIndex"]])
    assert (err.max(axis=1) <= 1).mean() > 0.9
    assert np.all(m["keypointIndex"] == -1)
    assert np.all(np.diff(m["featureIndex"]) > 0)  # prediction order
    assert np.all(m["distance"] <= 0.2 + 1e-6)     # zncc >= 0.8


def test_ncc_rejects_flat_and_unrelated_images(oracle_lib):
    seq = SyntheticSequence(30, 2)
    o = _seeded(seq)
    o.set_image(seq.render_image(0))
    o.capture_templates(np.arange(30), seq.pixel_positions(0).astype(np.float64))
    o.predict()
    preds, _, _ = o.predict_measurements()
    o.set_image(np.full((480, 640), 77, dtype=np.uint8))  # zero variance everywhere: no candidate scores
    assert len(o.match_ncc(preds)) == 0
    rng = np.random.default_rng(5)
    o.set_image(rng.integers(0, 256, (480, 640), dtype=np.uint8))
    assert len(o.match_ncc(preds)) == 0


def test_step_image_tracks(oracle_lib):
    seq = SyntheticSequence(50, 4)
    o = _seeded(seq)
    o.set_image(seq.render_image(0))
    o.capture_templates(np.arange(50), seq.pixel_positions(0).astype(np.float64))
    for t in range(1, 5):
        info = o.step_image(seq.render_image(t), ALGORITHMIC)
        assert info.status == 0
        assert info.n_matches >= 35
        assert info.n_inliers + info.n_rescued >= 0.8 * info.n_matches
    x = o.x13()
    assert np.linalg.norm(x[0:3] - seq.truth_r[4]) < 0.05


def test_detector_finds_unmasked_corners(oracle_lib):
    """new-feature detection (the build's own corner measure + the reference's zone heuristic)"""
    seq = SyntheticSequence(40, 2, horizon=2)  # points spread over the whole frame: every zone holds structure
    o = _seeded(seq)
    img = seq.render_image(0)
    o.set_image(img)
    none = np.zeros(0, dtype=[("featureIndex", "<i4"), ("_pad", "<i4"), ("imagePos", "<f8", (2,)),
                              ("covarianceMatrix", "<f8", (4,))])
    uv = o.detect_new_features(none, 25)
    assert len(uv) == 25
    # picks sit on the pasted texture patches (the only structure in the frame), away from the border
    px = seq.pixel_positions(0)
    d = np.abs(uv[:, None, :] - px[None, :, :]).max(axis=2).min(axis=1)
    assert (d <= 10).mean() > 0.9
    assert uv[:, 0].min() >= 16 and uv[:, 1].min() >= 16 and uv[:, 0].max() < 640 - 16 and uv[:, 1].max() < 480 - 16
    # spread: with 16 zones and 25 picks no zone that has candidates is left empty while another holds > 3
    zones = (uv[:, 1].astype(int) // 120) * 4 + uv[:, 0].astype(int) // 160
    assert np.bincount(zones, minlength=16).max() <= 4
    # mask of the picks themselves: no two closer than the disc radius (15 px)
    dd = np.sqrt(((uv[:, None, :] - uv[None, :, :]) ** 2).sum(-1)) + np.eye(len(uv)) * 1e9
    assert dd.min() > 15
    # predictions mask their ellipses: nothing is picked inside a predicted feature's gate
    o.predict()
    preds, _, _ = o.predict_measurements()
    uv2 = o.detect_new_features(preds, 25)
    for p in preds:
        ax, ang = o.ellipse(p["covarianceMatrix"])
        for q in uv2:
            assert not o.point_in_ellipse(float(q[0]), float(q[1]), float(p["imagePos"][0]), float(p["imagePos"][1]),
                                          int(np.rint(ax[0])), int(np.rint(ax[1])), ang)
