/*
 * ekf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, IEEE double) of the per-frame hot path of segeschecho/OpenEKFMonoSLAM:
 * predict -> measurement prediction + Jacobians -> ellipse-gated descriptor matching -> 1-point RANSAC ->
 * update -> outlier rescue -> second update.  It is the checker the HIP engine is compared against; nothing in
 * the product (openekfmonoslam_amd/, include/) links, loads or calls it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures and cannot be
 * built in this image (every translation unit includes OpenCV 2.4 headers, which are absent, and writing
 * stand-ins for them is not allowed).  This restatement therefore follows the reference source line by line
 * (citations below, relative to /root/reference/kalmanFilter/modules/; EKF/ = 1PointRansacEKF/) and restates
 * the pieces of OpenCV 2.4 the path calls (dense gemm, LU inverse, closed-form 2x2/3x3 inverse, Jacobi
 * eigen-decomposition) from their published algorithms.  The only reference-derived known answers available
 * are the ten matching-selection probe results recorded in SURVEY.md section 8(c); tests/golden/ holds them.
 */
#ifndef EKF_ORACLE_H
#define EKF_ORACLE_H

#include "../include/ekf_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcFilter OrcFilter;

/* update variants */
enum {
    ORC_UPDATE_LITERAL = 0,    /* dense stacked H, K = P H' inv(S) by LU, P = (I-KH) P : EKF/Update.cpp:92-109,214-218 */
    ORC_UPDATE_ALGORITHMIC = 1 /* same mathematics: A = H P (block-sparse), S by Cholesky, P -= A' inv(S) A       */
};

typedef struct OrcStepInfo {
    int32_t n_predicted;
    int32_t n_matches;
    int32_t n_hypotheses; /* RANSAC hypotheses evaluated                 */
    int32_t n_inliers;    /* low-innovation inliers (first update)       */
    int32_t n_outliers;   /* outlier matches re-predicted                */
    int32_t n_rescued;    /* high-innovation inliers (second update)     */
    int32_t status;
    int32_t _pad;
} OrcStepInfo;

OrcFilter *orc_create(const EkfCamera *cam, const EkfParams *par, int max_features);
void orc_destroy(OrcFilter *f);
int orc_set_descriptor_format(OrcFilter *f, int flags); /* EKF_DESCRIPTOR_* of ekf_engine.h; before any descriptor is stored */
int orc_descriptor_bytes(const OrcFilter *f);

/* initState + initCovariance: EKF/CommonFunctions.cpp:39-80.  Empties the map. */
void orc_reset(OrcFilter *f);

/* addFeatureToStateAndCovariance: EKF/AddMapFeature.cpp:293-344 (+ :109-289).  Returns new feature index or <0. */
int orc_add_feature(OrcFilter *f, const double uv[2], const uint8_t *desc32);

/* Map management, EKF/MapManagement.cpp: removeFeaturesFromStateAndCovariance :212 (ascending
 * feature indices), removeBadMapFeatures :279, computeLinearityIndex :312, convertToDepth :343,
 * convertMapFeaturesInverseDepthToDepth :494 (returns the converted feature index or -1). */
int orc_remove_features(OrcFilter *f, const int32_t *idx, int count);
int orc_remove_bad_features(OrcFilter *f);
double orc_linearity_index(const OrcFilter *f, int fi);
int orc_convert_to_depth(OrcFilter *f, int fi);
int orc_convert_inverse_depth_to_depth(OrcFilter *f);

/* Bulk load.  feature_pos holds 6 doubles per feature (depth features use the first 3); covariance positions
 * are assigned sequentially in map order (13, 13+d0, ...) as State/MapFeature do.  P is n x n row-major. */
int orc_set_state(OrcFilter *f, const double x13[13], int n_features, const double *feature_pos,
                  const int32_t *feature_type, const uint8_t *desc32, const double *P);

int orc_state_dim(const OrcFilter *f);    /* n = 13 + 3 d + 6 id */
int orc_num_features(const OrcFilter *f); /* N                   */
double *orc_x13(OrcFilter *f);            /* r(3) q(4: w x y z) v(3) w(3) */
double *orc_rotation(OrcFilter *f);       /* 3x3 row-major R(q), State::orientationRotationMatrix */
double *orc_feature_pos(OrcFilter *f);    /* 6 per feature */
int32_t *orc_feature_type(OrcFilter *f);
int32_t *orc_feature_covpos(OrcFilter *f);
uint8_t *orc_feature_desc(OrcFilter *f);
uint32_t *orc_feature_times_predicted(OrcFilter *f); /* MapFeature::timesPredicted */
uint32_t *orc_feature_times_matched(OrcFilter *f);   /* MapFeature::timesMatched   */
double *orc_P(OrcFilter *f); /* n x n row-major, leading dimension n */

/* stateAndCovariancePrediction: EKF/StateAndCovariancePrediction.cpp:244-253.  If F13/GQG13 are non-NULL
 * they receive the 13x13 F and G Q G' used. */
void orc_predict(OrcFilter *f, double *F13, double *GQG13);

/* predictMeasurementState: EKF/MeasurementPrediction.cpp:203-265, evaluated on an arbitrary camera state and
 * feature parameter array (so RANSAC hypotheses can reuse it).  feat_idx==NULL/count==0 means "all features".
 * Writes featureIndex + imagePos of every predicted feature (in input order); returns how many. */
int orc_predict_measurement_state(const OrcFilter *f, const double x13[13], const double R[9],
                                  const double *feature_pos, const int32_t *feat_idx, int count,
                                  EkfPrediction *out);

/* predictCameraMeasurements: EKF/MeasurementPrediction.cpp:705-719.  On return preds[k].covarianceMatrix
 * holds S_i; Hs (2x13 per prediction) and Hf (2x6 per prediction, first d columns used) hold the Jacobian
 * blocks; HP (optional, 2 x n per prediction) holds hiByP.  Returns number of predictions. */
int orc_predict_measurements(OrcFilter *f, const int32_t *feat_idx, int count, EkfPrediction *preds, double *Hs,
                             double *Hf, double *HP);

/* matrix2x2ToUncertaintyEllipse2D: Core/EKFMath.cpp:271-298.  axes = float semi-axes, angle in radians. */
void orc_ellipse(const double S[4], float axes[2], double *angle);
/* pointIsInsideEllipse: Core/EKFMath.cpp:302-351 (axes already rounded to int). */
int orc_point_in_ellipse(float px, float py, float cx, float cy, int ax_w, int ax_h, double angle);

/* The gate + distance + selection stage of matchPredictedFeatures: EKF/Matching.cpp:217-262, given the
 * keypoints/descriptors a detector produced.  Returns number of matches (prediction order). */
int orc_match(const OrcFilter *f, const EkfPrediction *preds, int n_pred, const EkfKeypoint *kps,
              const uint8_t *desc32, int n_kp, EkfMatch *out);

/* ransac: EKF/1PointRansac.cpp:101-234.  preds/Hs/Hf are index-aligned with matches.
 * inlier_mask[M] receives 0/1.  support_counts (optional, capacity M) receives the support size of each
 * evaluated hypothesis.  Returns number of hypotheses evaluated. */
int orc_ransac(OrcFilter *f, const EkfPrediction *preds, const double *Hs, const double *Hf,
               const EkfMatch *matches, int M, uint8_t *inlier_mask, int32_t *support_counts);

/* update: EKF/Update.cpp:282-319.  Index-aligned inputs, M matches.  Returns EKF_OK or an error. */
int orc_update(OrcFilter *f, const EkfMatch *matches, const EkfPrediction *preds, const double *Hs,
               const double *Hf, int M, int variant);

/* rescueOutliers: EKF/EKF.cpp:68-119.  rescued_mask[M] receives 0/1. Returns number rescued. */
int orc_rescue(const OrcFilter *f, const EkfMatch *matches, const EkfPrediction *preds, int M,
               uint8_t *rescued_mask);

/* One EKF::step (EKF/EKF.cpp:242-572) with a fixed map, fed the keypoints/descriptors of the frame; includes
 * updateMapFeatures (EKF/MapManagement.cpp:77-113: prediction/match counters, descriptor refresh of matched
 * features), which the reference runs every step. */
int orc_step(OrcFilter *f, const EkfKeypoint *kps, const uint8_t *desc32, int n_kp, int variant,
             OrcStepInfo *info);

/* NCC matcher (mode B: 11x11 ZNCC inside the predicted ellipse, 3-level pyramid, coarse-to-fine).  No reference
 * counterpart -- the build's own definition (see ekf_oracle.c); one current image per process. */
int orc_image_set(const uint8_t *image, int w, int h, int stride, int channels);
const uint8_t *orc_image_level(int level, int *w, int *h);
int orc_capture_templates(OrcFilter *f, const int32_t *feat_idx, const double *uv, int count);
const uint8_t *orc_templates(void);
int orc_match_ncc(const OrcFilter *f, const EkfPrediction *preds, int n_pred, EkfMatch *out);
int orc_step_image(OrcFilter *f, int variant, OrcStepInfo *info);
/* detectNewImageFeatures (EKF/DetectNewImageFeatures.cpp:337) on the current image, the build's own detector; returns count */
int orc_detect_new_features(const EkfPrediction *preds, int n_pred, int max_new, int divide_times,
                            double mask_ellipse_size, double min_response, double *uv_out);

/* Timing helper for bench.py's cpu_baseline: runs the three dense products of the LITERAL covariance update
 * (K H, I - K H, (I - K H) P ; EKF/Update.cpp:214-218) restricted to the first `rows` rows of the result and
 * returns the wall seconds spent.  K is n x m, H is m x n, P is n x n, all random-filled by the caller. */
double orc_time_literal_rows(int n, int m, int rows, const double *K, const double *H, const double *P,
                             double *out_rows);

#ifdef __cplusplus
}
#endif
#endif
