"""ctypes / numpy mirrors of include/ekf_types.h (plain-old-data crossing the C ABI).

Field names follow the reference's own classes (CameraCalibration.h:37-63,
ExtendedKalmanFilterParameters.h:37-76, ImageFeaturePrediction.h:37-50, Matching.h:38-48).
"""
import ctypes as C

import numpy as np


class EkfCamera(C.Structure):
    _fields_ = [
        ("pixelsX", C.c_int32),
        ("pixelsY", C.c_int32),
        ("fx", C.c_double),
        ("fy", C.c_double),
        ("k1", C.c_double),
        ("k2", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
        ("dx", C.c_double),
        ("dy", C.c_double),
        ("pixelErrorX", C.c_double),
        ("pixelErrorY", C.c_double),
        ("angularVisionX", C.c_double),
        ("angularVisionY", C.c_double),
    ]


class EkfParams(C.Structure):
    _fields_ = [
        ("initInvDepthRho", C.c_double),
        ("initLinearAccelSD", C.c_double),
        ("initAngularAccelSD", C.c_double),
        ("linearAccelSD", C.c_double),
        ("angularAccelSD", C.c_double),
        ("inverseDepthRhoSD", C.c_double),
        ("matchingCompCoefSecondBestVSFirst", C.c_double),
        ("ransacThresholdPredictDistance", C.c_double),
        ("ransacAllInliersProbability", C.c_double),
        ("ransacChi2Threshold", C.c_double),
        ("goodFeatureMatchingPercent", C.c_double),
        ("inverseDepthLinearityIndexThreshold", C.c_double),
    ]


class EkfStepInfo(C.Structure):
    """OrcStepInfo (oracle) and EkfStepInfo (engine) share this layout."""

    _fields_ = [
        ("n_predicted", C.c_int32),
        ("n_matches", C.c_int32),
        ("n_hypotheses", C.c_int32),
        ("n_inliers", C.c_int32),
        ("n_outliers", C.c_int32),
        ("n_rescued", C.c_int32),
        ("status", C.c_int32),
        ("n_sweep_retries", C.c_int32),
    ]


PREDICTION_DTYPE = np.dtype(
    [("featureIndex", "<i4"), ("_pad", "<i4"), ("imagePos", "<f8", (2,)), ("covarianceMatrix", "<f8", (4,))]
)
MATCH_DTYPE = np.dtype(
    [("featureIndex", "<i4"), ("keypointIndex", "<i4"), ("imagePos", "<f8", (2,)), ("distance", "<f4"), ("_pad", "<f4")]
)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4")])
assert PREDICTION_DTYPE.itemsize == 56 and MATCH_DTYPE.itemsize == 32 and KEYPOINT_DTYPE.itemsize == 8

DESC_BYTES = 32
FEATURE_DEPTH = 1
FEATURE_INVERSE_DEPTH = 2

EKF_OK = 0
STATUS_NAMES = {
    0: "EKF_OK",
    1: "EKF_ERR_INVALID_ARG",
    2: "EKF_ERR_CAPACITY",
    3: "EKF_ERR_NOT_POSITIVE_DEFINITE",
    4: "EKF_ERR_NON_FINITE",
    5: "EKF_ERR_HIP",
    6: "EKF_ERR_NO_DEVICE",
    7: "EKF_ERR_COMM",
    8: "EKF_ERR_TIMEOUT",
}


def s3_camera(width=640, height=480):
    """Camera "S3" of the reference's shipped experiment (experiments/s3/config.yml:49-63), scaled to the
    frame size the way SURVEY.md section 8(d) prescribes: fx fy cx cy scale with width/640, the pixel pitch
    dx dy with its inverse, k1 k2 and the field of view are kept."""
    s = width / 640.0
    cam = EkfCamera()
    cam.pixelsX, cam.pixelsY = int(width), int(height)
    cam.fx = 525.060143149240389 * s
    cam.fy = 524.245488213640215 * s
    cam.k1 = -7.613e-003
    cam.k2 = 9.388e-004
    cam.cx = 308.649343121753361 * s
    cam.cy = 236.536005491807288 * s
    cam.dx = 0.007021618750000 / s
    cam.dy = 0.007027222916667 / s
    cam.pixelErrorX = 1.0
    cam.pixelErrorY = 1.0
    cam.angularVisionX = 62.720770890650357
    cam.angularVisionY = 49.163954709609868
    return cam


def s3_params():
    """EKF profile "EKF" of experiments/s3/config.yml:10-37."""
    p = EkfParams()
    p.initInvDepthRho = 1.0
    p.initLinearAccelSD = 0.001
    p.initAngularAccelSD = 0.004
    p.linearAccelSD = 0.0007
    p.angularAccelSD = 0.002
    p.inverseDepthRhoSD = 1.0
    p.matchingCompCoefSecondBestVSFirst = 1.0
    p.ransacThresholdPredictDistance = 1.0
    p.ransacAllInliersProbability = 0.99
    p.ransacChi2Threshold = 5.9915
    p.goodFeatureMatchingPercent = 0.5
    p.inverseDepthLinearityIndexThreshold = 0.1
    return p
