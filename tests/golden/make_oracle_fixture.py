"""Mints tests/golden/oracle_n12_3frames.npz: inputs (the seeded N = 12 synthetic sequence) and the oracle's outputs
after each of 3 frames (LITERAL update variant).  These are ORACLE-minted vectors (the reference cannot be built here,
see DESIGN.md section 2): they pin the oracle against drift and let the GPU tests check the engine against committed
numbers.  Run from the repository root:  python tests/golden/make_oracle_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402


def main():
    seq = SyntheticSequence(12, 3)
    o = ol.Oracle(seq.cam, seq.par, 16)
    o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    out = {"x13_0": seq.x13, "feature_pos_0": seq.feature_pos, "feature_desc": seq.feature_desc, "P_0": seq.P0}
    for t, (kps, desc) in enumerate(seq.frames):
        info = o.step(kps, desc, ol.LITERAL)
        out[f"kps_{t}"] = np.stack([kps["x"], kps["y"]], -1)
        out[f"desc_{t}"] = desc
        out[f"info_{t}"] = np.array([info.n_predicted, info.n_matches, info.n_hypotheses, info.n_inliers, info.n_outliers,
                                     info.n_rescued, info.status], dtype=np.int32)
        out[f"x13_{t + 1}"] = o.x13()
        out[f"feature_pos_{t + 1}"] = o.feature_pos()
        out[f"P_{t + 1}"] = o.P()
    path = os.path.join(ROOT, "tests", "golden", "oracle_n12_3frames.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
