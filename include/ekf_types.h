/*
 * ekf_types.h -- plain-old-data types shared by the C ABI (ekf_engine.h), the C++ compat layer and the
 * test oracle.  C99 / C++ compatible, no dependencies.
 *
 * Reference citations are relative to /root/reference/kalmanFilter/modules/ :
 *   EKF/ = 1PointRansacEKF/, Cfg/ = Configuration/ConfigurationDataReader/.
 */
#ifndef EKF_TYPES_H
#define EKF_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Camera intrinsics consumed on the hot path.
 * Mirrors class CameraCalibration, Cfg/CameraCalibrationConfiguration/CameraCalibration.h:37-63
 * (same field names, same meaning; the std::string name is dropped). */
typedef struct EkfCamera {
    int32_t pixelsX;
    int32_t pixelsY;
    double fx, fy;
    double k1, k2;
    double cx, cy;
    double dx, dy;
    double pixelErrorX, pixelErrorY;
    double angularVisionX, angularVisionY; /* degrees */
} EkfCamera;

/* Filter parameters consumed on the hot path.
 * Mirrors class ExtendedKalmanFilterParameters,
 * Cfg/ExtendedKalmanFilterConfiguration/ExtendedKalmanFilterParameters.h:37-76
 * (only the fields read by stages A1..A9 of SURVEY.md section 8(a), plus those the map-management row
 * 8(f)-1 reads: new-feature initialisation, bad-feature removal, inverse-depth -> depth conversion). */
typedef struct EkfParams {
    double initInvDepthRho;
    double initLinearAccelSD;
    double initAngularAccelSD;
    double linearAccelSD;
    double angularAccelSD;
    double inverseDepthRhoSD;
    double matchingCompCoefSecondBestVSFirst;
    double ransacThresholdPredictDistance;
    double ransacAllInliersProbability;
    double ransacChi2Threshold;
    double goodFeatureMatchingPercent;          /* removeBadMapFeatures, EKF/MapManagement.cpp:279-308        */
    double inverseDepthLinearityIndexThreshold; /* convertMapFeaturesInverseDepthToDepth, MapManagement.cpp:494 */
} EkfParams;

/* Map feature parametrisation, enum MapFeatureType EKF/MapFeature.h:39-44 (same numeric values). */
enum {
    EKF_FEATURE_INVALID = 0,
    EKF_FEATURE_DEPTH = 1,        /* 3 parameters: x y z                 */
    EKF_FEATURE_INVERSE_DEPTH = 2 /* 6 parameters: x y z theta phi rho   */
};

/* Binary descriptor width used by the reference's configured extractor (BRIEF-32,
 * Cfg/DescriptorExtractorConfiguration/DescriptorExtractorFactory.cpp:114-123). */
#define EKF_DESC_BYTES 32

/* A predicted measurement: class ImageFeaturePrediction, EKF/ImageFeaturePrediction.h:37-50. */
typedef struct EkfPrediction {
    int32_t featureIndex;
    int32_t _pad;
    double imagePos[2];
    double covarianceMatrix[4]; /* 2x2 row-major innovation covariance S_i (R_i = I) */
} EkfPrediction;

/* A match: class FeatureMatch, EKF/Matching.h:38-48 (descriptor omitted; keypointIndex added so a
 * caller can fetch it from its own descriptor matrix). */
typedef struct EkfMatch {
    int32_t featureIndex;
    int32_t keypointIndex;
    double imagePos[2];
    float distance;
    float _pad;
} EkfMatch;

/* A detected keypoint position: cv::KeyPoint::pt (float x, y) as read at EKF/Matching.cpp:232,250-256. */
typedef struct EkfKeypoint {
    float x, y;
} EkfKeypoint;

/* Numeric constants of the reference, Core/EKFMath.h:37-41 (long double literals there; used as double). */
#define EKF_EPSILON 2.22e-16
#define EKF_DELTA 1.0e-12
#define EKF_PI 3.14159265
#define EKF_CHISQ_95_2 5.9915

/* Status codes returned across the C ABI. */
enum {
    EKF_OK = 0,
    EKF_ERR_INVALID_ARG = 1,
    EKF_ERR_CAPACITY = 2,
    EKF_ERR_NOT_POSITIVE_DEFINITE = 3, /* S = H P H' + R failed to factorise (reference: silent zeros from cv::invert) */
    EKF_ERR_NON_FINITE = 4,
    EKF_ERR_HIP = 5,
    EKF_ERR_NO_DEVICE = 6,
    EKF_ERR_COMM = 7,
    EKF_ERR_TIMEOUT = 8 /* a device-side wait of the persistent Cholesky sweep exceeded its bound (its workgroups were not all resident:
                           another persistent launch of another process on the same GPU?); the update is incomplete */
};

#ifdef __cplusplus
}
#endif
#endif /* EKF_TYPES_H */
