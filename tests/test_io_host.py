"""Host pieces either side of the path, CPU only: the reference's config.yml schema, PNG
sequences, and the output.yml layout -- openekfmonoslam_amd/compat/ekf_io.h driven through tests/cpp/io_check.cpp."""
import os
import re
import subprocess

import numpy as np
import pytest
import yaml

from openekfmonoslam_amd.ekftypes import s3_camera, s3_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the reference's schema (experiments/s3/config.yml) with the S3 values of SURVEY.md appendix A; written by the test
CONFIG = """%%YAML:1.0

RunConfiguration:
  ExtendedKalmanFilter: "EKF"
  FeatureDetector: "STAR"
  DescriptorExtractor: "BRIEF"
  CameraCalibration: "S3"

ExtendedKalmanFilter:
  Other:
    InitInvDepthRho: "7.0"
  EKF:
    ReserveFeaturesDepth: "1024"
    ReserveFeaturesInvDepth: "512"
    InitInvDepthRho: "1.0"
    InitLinearAccelSD: "0.001"
    InitAngularAccelSD: "0.004"
    LinearAccelSD: "0.0007"
    AngularAccelSD: "0.002"
    InverseDepthRhoSD: "1.0"
    MaxMapSize: "240"
    AlwaysRemoveUnseenMapFeatures: "true"
    MapManagementFrequency: "1"
    DetectNewFeaturesImageAreasDivideTimes: "2"
    DetectNewFeaturesImageMaskEllipseSize: "10"
    MatchingCompCoefSecondBestVSFirst: "1.0"
    MinMatchesPerImage: "%(min_matches)d"
    GoodFeatureMatchingPercent: "0.5"
    RansacThresholdPredictDistance: "1.0"
    RansacAllInliersProbability: "0.99"
    RansacChi2Threshold: "5.9915"
    InverseDepthLinearityIndexThreshold: "0.1"

FeatureDetector:
  STAR:
    Type: "STAR"

DescriptorExtractor:
  BRIEF:
    Type: "BRIEF"

CameraCalibration:
  S3:
    PixelsX: "640"
    PixelsY: "480"
    FX: "525.060143149240389"
    FY: "524.245488213640215"
    K1: "-7.613e-003"
    K2: "9.388e-004"
    CX: "308.649343121753361"
    CY: "236.536005491807288"
    DX: "0.007021618750000"
    DY: "0.007027222916667"
    PixelErrorX: "1.0"
    PixelErrorY: "1.0"
    AngularVisionX: "62.720770890650357"
    AngularVisionY: "49.163954709609868"
"""


@pytest.fixture(scope="module")
def io_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("io") / "io_check")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-o", exe, os.path.join(ROOT, "tests", "cpp", "io_check.cpp"), "-lz"])
    return exe


def _parse_config_output(out):
    rows = {ln.split()[0]: [float(v) for v in ln.split()[1:]] for ln in out.strip().splitlines()}
    return rows["cam"], rows["par"], rows["run"]


def _expect(cam, par):
    c, p = s3_camera(640, 480), s3_params()
    np.testing.assert_array_equal(cam, [c.pixelsX, c.pixelsY, c.fx, c.fy, c.k1, c.k2, c.cx, c.cy, c.dx, c.dy, c.pixelErrorX,
                                        c.pixelErrorY, c.angularVisionX, c.angularVisionY])
    np.testing.assert_array_equal(par, [p.initInvDepthRho, p.initLinearAccelSD, p.initAngularAccelSD, p.linearAccelSD,
                                        p.angularAccelSD, p.inverseDepthRhoSD, p.matchingCompCoefSecondBestVSFirst,
                                        p.ransacThresholdPredictDistance, p.ransacAllInliersProbability,
                                        p.ransacChi2Threshold, p.goodFeatureMatchingPercent,
                                        p.inverseDepthLinearityIndexThreshold])


def test_config_loader_reads_the_reference_schema(io_check, tmp_path):
    cfg = tmp_path / "config.yml"
    cfg.write_text(CONFIG % {"min_matches": 60})
    cam, par, run = _parse_config_output(subprocess.check_output([io_check, "config", str(cfg)], text=True))
    _expect(cam, par)
    assert run == [1024, 512, 240, 0, 1, 1, 2, 10.0, 60]
    # a missing profile is an error, not a silent default
    bad = tmp_path / "bad.yml"
    bad.write_text((CONFIG % {"min_matches": 60}).replace('CameraCalibration: "S3"', 'CameraCalibration: "Nope"'))
    r = subprocess.run([io_check, "config", str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "not found" in r.stdout


def test_config_loader_on_the_reference_file_itself(io_check):
    ref = "/root/reference/experiments/s3/config.yml"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    cam, par, run = _parse_config_output(subprocess.check_output([io_check, "config", ref], text=True))
    _expect(cam, par)
    assert run[8] == 60 and run[4] == 1


def test_png_reader_and_writer_against_pillow(io_check, tmp_path):
    from PIL import Image

    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    rgb[5:20, 7:30] = np.linspace(0, 255, 23, dtype=np.uint8)[None, :, None]  # smooth area: exercises the row filters
    cases = {"rgb": Image.fromarray(rgb), "gray": Image.fromarray(rgb[..., 0]), "rgba": Image.fromarray(np.dstack([rgb, rgb[..., :1]])),
             "pal": Image.fromarray(rgb).quantize(64)}
    for name, im in cases.items():
        src, dst = str(tmp_path / f"{name}.png"), str(tmp_path / f"{name}_out.png")
        im.save(src, optimize=True)
        out = subprocess.check_output([io_check, "png", src, dst], text=True).split()
        w, h, ch, s = int(out[1]), int(out[2]), int(out[3]), int(out[4])
        ref = np.asarray(im.convert("L") if name == "gray" else im.convert("RGB"))
        assert (w, h) == (53, 37) and ch == (1 if name == "gray" else 3)
        flat = (ref if ch == 1 else ref[..., ::-1]).reshape(-1).astype(np.uint64)   # reader delivers B G R
        assert s == int((flat * (np.arange(flat.size, dtype=np.uint64) % 251 + 1)).sum())
        back = np.asarray(Image.open(dst).convert("L" if ch == 1 else "RGB"))
        np.testing.assert_array_equal(back, ref)


def test_output_yml_layout(io_check, tmp_path):
    path = tmp_path / "output.yml"
    subprocess.check_call([io_check, "yaml", str(path)])
    text = path.read_text()
    assert text.startswith("%YAML:1.0\n")
    assert '"Frame 1":' in text and "!!opencv-matrix" in text and "# Running time (microseconds)" in text
    body = re.sub(r"!!opencv-matrix", "", text.split("\n", 1)[1])
    doc = yaml.safe_load(body)
    f1 = doc["Frame 2"]
    assert f1["totalMatches"] == 58 and f1["MapFeaturesInvDepthCount"] == 40
    assert f1["Prediction"] == pytest.approx(124.456, abs=1e-12) and f1["UpdateLI"] == 2000.0
    st = f1["StateEstimation"]
    assert (st["rows"], st["cols"], st["dt"]) == (1, 13, "d")
    np.testing.assert_allclose(st["data"], [-(i + 1) / 7.0 for i in range(13)], rtol=1e-15)
    P = np.array(doc["Frame 1"]["StateCovarianceMatrixEstimation"]["data"]).reshape(13, 13)
    np.testing.assert_allclose(np.diag(P), [1e-6 * (14 * i + 1) for i in range(13)], rtol=1e-15)
    assert all(len(ln) <= 80 for ln in text.splitlines())


def test_prediction_image_drawing(io_check, tmp_path):
    """drawPrediction (Gui/Draw.cpp:266-310): cross + ellipse outline + index per prediction, red for inverse-depth and
    green for depth features, on a BGR copy of the frame."""
    from PIL import Image

    path = tmp_path / "pred.png"
    subprocess.check_call([io_check, "draw", str(path)])
    im = np.asarray(Image.open(str(path)).convert("RGB")).astype(int)
    assert im.shape == (64, 96, 3)
    red, green = np.array([255, 0, 0]), np.array([0, 255, 0])
    # crosses: centre pixel and arm ends at (int)u, (int)v (the right arm lies under the index label, drawn last)
    for (x, y), col in (((30, 20), red), ((70, 40), green)):
        for dx, dy in ((0, 0), (-2, 0), (0, 2), (0, -2)):
            assert (im[y + dy, x + dx] == col).all(), (x, y, dx, dy)
    # ellipse of S = diag(4, 1): semi-axes (int)(2 sqrt(4 * 5.9915)) = 9 and (int)(2 sqrt(5.9915)) = 4 around (30, 20)
    assert (im[20, 30 + 9] == red).all() and (im[20, 30 - 9] == red).all()
    assert (im[20 + 4, 30] == red).all() and (im[20 - 4, 30] == red).all()
    assert (im[20, 30 + 11] == 40).all()  # outside the outline: untouched background
    # circle of S = I: radius 4 around (70, 40), green
    assert (im[40, 66] == green).all() and (im[44, 70] == green).all()
    # index labels: yellow text (BGR 0,255,255 = RGB 255,255,0) with a gray shadow
    assert ((im == [255, 255, 0]).all(axis=2)).sum() > 10 and ((im == [150, 150, 150]).all(axis=2)).sum() > 0


def test_log_txt_layout(io_check, tmp_path):
    """State::showDetailed (State.cpp:229-258, 371-400; MapFeature.cpp:130-145), default ostream formatting."""
    path = tmp_path / "log.txt"
    subprocess.check_call([io_check, "log", str(path)])
    lines = path.read_text().splitlines()
    assert lines[0] == "Posicion de la camara: 0.5, -0.25, 1"
    assert lines[1] == "Orientacion(cuaternions): 1, 0, 0, 0"
    assert lines[2] == "Orientacion en angulos eulerianos: 0, 0, 0"
    assert lines[3] == "Velocidad lineal (con respecto al mundo): 0.01, 0, 0.002"
    assert lines[4] == "Velocidad angular (con respecto a la camara): 0, 0.002, 0"
    assert lines[5] == "Cantidad de features en el mapa: 2"
    assert lines[6] == "" and lines[7] == "Map Features (2):"
    assert lines[8] == "0: 0, 0, 0, 0.1, -0.2, 0.5 (5/7)"
    assert lines[9] == "1: 1.5, 2.5, 3.5 (3/3)"
