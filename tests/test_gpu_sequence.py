"""The reference's sample command line on the engine (samples/ekf_sequence.cpp):
`ekf_sequence config.yml imgdir/ outdir/` on a rendered PNG sequence -- init on frame 0 (detect + add + templates),
steps with device map management, output.yml in the reference's layout."""
import os
import re
import subprocess

import numpy as np
import pytest
import yaml

from openekfmonoslam_amd.synth import SyntheticSequence
from tests.test_io_host import CONFIG

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "openekfmonoslam_amd")


def test_sample_program_runs_a_png_sequence(tmp_path):
    from PIL import Image

    exe = str(tmp_path / "ekf_sequence")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-o", exe, os.path.join(ROOT, "samples", "ekf_sequence.cpp"),
                           "-L", PKG, "-lekf_engine", "-lz", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"])
    seq = SyntheticSequence(80, 6)
    imgdir, outdir = tmp_path / "img", tmp_path / "out"
    imgdir.mkdir()
    outdir.mkdir()
    for t in range(7):
        frame = seq.render_image(t, outlier_fraction=0.0, channels=3 if t % 2 == 0 else 1)
        Image.fromarray(frame[..., ::-1] if frame.ndim == 3 else frame).save(str(imgdir / f"{t:05d}.png"))
    cfg = tmp_path / "config.yml"
    cfg.write_text(CONFIG % {"min_matches": 40})
    r = subprocess.run([exe, str(cfg), str(imgdir) + "/", str(outdir) + "/"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "init: 40 features"
    steps = [ln for ln in lines if ln.startswith("step")]
    assert len(steps) == 6
    for ln in steps:
        m = re.match(r"step (\d+): predicted (\d+) matches (\d+) li (\d+) hi (\d+) features (\d+)", ln)
        pred, matches, li, hi, feats = (int(m.group(i)) for i in range(2, 7))
        assert matches >= 0.7 * pred and li + hi >= 0.8 * matches
        assert feats >= 40 - 5  # map management tops the map up towards MinMatchesPerImage
    text = (outdir / "output.yml").read_text()
    doc = yaml.safe_load(re.sub(r"!!opencv-matrix", "", text.split("\n", 1)[1]))
    assert sorted(doc) == [f"Frame {k}" for k in range(1, 7)]  # EKF.cpp:244: the first step is Frame 1
    keys = ["Prediction", "Matching", "Ransac", "totalMatches", "liInliers", "UpdateLI", "RescueOutliers", "hiInliers",
            "UpdateHI", "MapManagement", "StateEstimation", "MapFeaturesInvDepthCount", "MapFeaturesDepthCount",
            "StateCovarianceMatrixEstimation"]
    last = doc["Frame 6"]
    assert list(last) == keys  # same keys, same order as EKF.cpp:262-628
    assert last["UpdateLI"] > 0 and last["Prediction"] > 0
    x = np.array(last["StateEstimation"]["data"])
    printed = [float(v) for v in steps[-1].split("r =")[1].split()]
    np.testing.assert_allclose(x[:3], printed, atol=1e-6)
    assert abs(np.linalg.norm(x[3:7]) - 1.0) < 1e-9
    P = np.array(last["StateCovarianceMatrixEstimation"]["data"]).reshape(13, 13)
    np.testing.assert_array_equal(P, P.T)
    assert np.all(np.diag(P) >= 0)
    # log.txt (EKF.cpp:135, 172-180, 233-236, 246-250, 662-665) and the per-frame prediction images (:294-305)
    log = (outdir / "log.txt").read_text()
    assert log.startswith("Random Seed: 0\n\n~~~~~~~~~~~~ STEP 0 ~~~~~~~~~~~~\n")
    assert log.count("~~~~~~~~~~~~ STEP ") == 7 and log.count("Map Features (") == 7
    assert "Cantidad de features en el mapa: 40" in log
    for k in range(1, 7):
        im = np.asarray(Image.open(str(outdir / f"{k:05d}.png")).convert("RGB")).astype(int)
        assert im.shape == (seq.cam.pixelsY, seq.cam.pixelsX, 3)
        assert ((im == [255, 0, 0]).all(axis=2)).sum() > 40 * 9  # a red cross (+ ellipse) per predicted inverse-depth feature
