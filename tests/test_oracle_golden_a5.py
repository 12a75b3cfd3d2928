"""A5 selection stage against the reference's own recorded answers (tests/golden/a5_selection_cases.json)."""
import json
import os

import numpy as np

from openekfmonoslam_amd.ekftypes import DESC_BYTES, KEYPOINT_DTYPE, PREDICTION_DTYPE, s3_camera, s3_params

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "a5_selection_cases.json")


def desc_with_distance(d):
    """32-byte descriptor at Hamming distance d from the all-zero map descriptor."""
    out = np.zeros(DESC_BYTES, dtype=np.uint8)
    for b in range(d):
        out[b // 8] |= np.uint8(1 << (b % 8))
    return out


def load_cases():
    with open(GOLDEN) as f:
        g = json.load(f)
    out = []
    for c in g["cases"]:
        kps = np.zeros(len(c["kps"]), dtype=KEYPOINT_DTYPE)
        kps["x"] = [p[0] for p in c["kps"]]
        kps["y"] = [p[1] for p in c["kps"]]
        desc = np.stack([desc_with_distance(d) for d in c["dists"]])
        out.append((c["name"], kps, desc, c["expect"]))
    pred = np.zeros(1, dtype=PREDICTION_DTYPE)
    pred["featureIndex"] = 0
    pred["imagePos"][0] = g["center"]
    pred["covarianceMatrix"][0] = g["S"]
    return pred, out


def test_a5_selection_cases_oracle(oracle_lib):
    cam, par = s3_camera(), s3_params()
    o = oracle_lib.Oracle(cam, par, 4)
    x = np.zeros(13)
    x[3] = 1
    o.set_state(x, np.zeros((1, 6)), None, np.zeros((1, DESC_BYTES), np.uint8), np.eye(19))
    pred, cases = load_cases()
    assert len(cases) == 9
    for name, kps, desc, expect in cases:
        m = o.match(pred, kps, desc)
        got = int(m["keypointIndex"][0]) if len(m) else -1
        assert got == expect, name
        if expect >= 0:
            assert m["imagePos"][0][0] == np.float64(kps["x"][expect]) and m["featureIndex"][0] == 0
    # empty keypoint set: no match
    assert len(o.match(pred, np.zeros(0, dtype=KEYPOINT_DTYPE), np.zeros((0, DESC_BYTES), np.uint8))) == 0


def test_a5_selection_cases_oracle_l2_branch(oracle_lib):
    """computeDistance's CV_32F branch (Matching.cpp:60-73) feeds the SAME list logic: float descriptors whose L2
    distances to the map descriptor are the recorded Hamming values must select the same keypoints."""
    cam, par = s3_camera(), s3_params()
    cols = 16
    o = oracle_lib.Oracle(cam, par, 4, descriptor_cols_f32=cols)
    x = np.zeros(13)
    x[3] = 1
    o.set_state(x, np.zeros((1, 6)), None, np.zeros((1, cols), np.float32), np.eye(19))
    pred, cases = load_cases()
    with open(GOLDEN) as f:
        dists = [c["dists"] for c in json.load(f)["cases"]]
    for (name, kps, _, expect), dd in zip(cases, dists):
        desc = np.zeros((len(dd), cols), np.float32)
        desc[:, 3] = dd  # |desc - 0| = d exactly
        m = o.match(pred, kps, desc)
        got = int(m["keypointIndex"][0]) if len(m) else -1
        assert got == expect, name
        if expect >= 0:
            assert m["distance"][0] == np.float32(dd[expect])


def test_l2_distance_arithmetic_matches_the_reference_expression(oracle_lib):
    """float difference, float square, double accumulation in column order, sqrt, stored as float (DMatch)."""
    cam, par = s3_camera(), s3_params()
    cols = 64
    rng = np.random.default_rng(5)
    q = rng.standard_normal(cols).astype(np.float32)
    c = rng.standard_normal((1, cols)).astype(np.float32)
    o = oracle_lib.Oracle(cam, par, 4, descriptor_cols_f32=cols)
    x = np.zeros(13)
    x[3] = 1
    o.set_state(x, np.zeros((1, 6)), None, q.reshape(1, cols), np.eye(19))
    pred, _ = load_cases()
    kps = np.zeros(1, dtype=KEYPOINT_DTYPE)
    kps["x"], kps["y"] = 100, 100
    m = o.match(pred, kps, c)
    acc = 0.0
    for j in range(cols):
        subs = np.float32(q[j] - c[0, j])
        acc += float(np.float32(subs * subs))
    assert len(m) == 1 and m["distance"][0] == np.float32(np.sqrt(acc))
