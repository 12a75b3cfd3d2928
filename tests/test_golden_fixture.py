"""Committed oracle-minted vectors (tests/golden/oracle_n12_3frames.npz, made by tests/golden/make_oracle_fixture.py):
the oracle must still reproduce them (CPU), and the HIP engine must agree with them (GPU) -- without the oracle in the
loop on the GPU side."""
import os

import numpy as np
import pytest

from openekfmonoslam_amd.ekftypes import KEYPOINT_DTYPE
from openekfmonoslam_amd.synth import SyntheticSequence

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_n12_3frames.npz")


def _frames(z):
    out = []
    for t in range(3):
        kps = np.zeros(len(z[f"kps_{t}"]), dtype=KEYPOINT_DTYPE)
        kps["x"], kps["y"] = z[f"kps_{t}"][:, 0], z[f"kps_{t}"][:, 1]
        out.append((kps, z[f"desc_{t}"]))
    return out


def _info(i):
    return [i.n_predicted, i.n_matches, i.n_hypotheses, i.n_inliers, i.n_outliers, i.n_rescued, i.status]


def test_generator_and_oracle_still_reproduce_the_fixture(oracle_lib):
    z = np.load(FIX)
    seq = SyntheticSequence(12, 3)
    np.testing.assert_array_equal(seq.P0, z["P_0"])  # the synthetic generator is part of the contract
    o = oracle_lib.Oracle(seq.cam, seq.par, 16)
    ftype = np.full(12, 2, dtype=np.int32)
    o.set_state(z["x13_0"], z["feature_pos_0"], ftype, z["feature_desc"], z["P_0"])
    for t, (kps, desc) in enumerate(_frames(z)):
        info = o.step(kps, desc, oracle_lib.LITERAL)
        assert _info(info) == list(z[f"info_{t}"])
        np.testing.assert_allclose(o.x13(), z[f"x13_{t + 1}"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(o.P(), z[f"P_{t + 1}"], rtol=1e-11, atol=1e-18)


@pytest.mark.gpu
def test_engine_against_the_committed_vectors():
    from openekfmonoslam_amd import engine

    z = np.load(FIX)
    seq = SyntheticSequence(12, 3)
    e = engine.EkfEngine(seq.cam, seq.par, 16, max_keypoints=128)
    e.set_state(z["x13_0"], z["feature_pos_0"], np.full(12, 2, dtype=np.int32), z["feature_desc"], z["P_0"])
    for t, (kps, desc) in enumerate(_frames(z)):
        info = e.step(kps, desc)
        assert _info(info) == list(z[f"info_{t}"])
        x, fp, P = e.get_state()
        assert np.abs(x - z[f"x13_{t + 1}"]).max() <= 1e-9
        assert np.abs(fp - z[f"feature_pos_{t + 1}"]).max() <= 1e-8
        Pr = z[f"P_{t + 1}"]
        assert np.abs(P - Pr).max() / np.abs(Pr).max() <= 1e-9
