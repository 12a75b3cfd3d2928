"""BASELINE configs[0] ("samples/ monocular sequence, plumbing, no GPU"): the image-in pipeline definition (own detector
+ NCC templates + the filter, i.e. exactly what ekf_compat::ImageEKF drives on the GPU) run by the CPU oracle on the
reference's own sample frames, experiments/s3/costado_recto1/%05d.png from frame 90 on, as samples/EKF/main.cpp does.
Reads /root/reference, so it only runs in the build container (skipped on the GPU box); no parity claim, it shows that
the definitions the HIP kernels are bit-checked against behave on real imagery."""
import os

import numpy as np
import pytest

from openekfmonoslam_amd.ekftypes import PREDICTION_DTYPE, s3_camera, s3_params
from tests.oracle_lib import ALGORITHMIC

SEQ = "/root/reference/experiments/s3/costado_recto1"


@pytest.mark.skipif(not os.path.isdir(SEQ), reason="reference sample frames not present")
def test_oracle_tracks_the_reference_sample_sequence(oracle_lib):
    from PIL import Image

    def load(i):  # B G R like cv::imread (FileSequenceImageGenerator.cpp:82)
        return np.asarray(Image.open(f"{SEQ}/{i:05d}.png").convert("RGB"))[..., ::-1].copy()

    o = oracle_lib.Oracle(s3_camera(640, 480), s3_params(), 128)
    o.reset()
    o.set_image(load(90))
    uv = o.detect_new_features(np.zeros(0, dtype=PREDICTION_DTYPE), 60, min_response=1e10)  # MinMatchesPerImage = 60
    assert len(uv) == 60
    for p in uv:
        o.add_feature(p)
    o.capture_templates(np.arange(60), uv)
    xs = []
    for t in range(91, 99):
        info = o.step_image(load(t), ALGORITHMIC)
        assert info.status == 0
        assert info.n_predicted == 60 and info.n_matches >= 50
        assert info.n_inliers + info.n_rescued >= 0.9 * info.n_matches
        xs.append(o.x13())
    xs = np.array(xs)
    assert np.all(np.isfinite(xs)) and np.allclose(np.linalg.norm(xs[:, 3:7], axis=1), 1.0, atol=1e-9)
    # "costado recto": the camera slides sideways at a steady pace -- x decreases monotonically, y and z stay small
    assert np.all(np.diff(xs[:, 0]) < 0)
    assert np.abs(xs[:, 1:3]).max() < 0.3 * abs(xs[-1, 0])
