/*
 * ekf_engine.h -- C ABI of the MI355X-native EKF engine (libekf_engine.so).
 *
 * Drop-in boundary for the per-frame hot path of segeschecho/OpenEKFMonoSLAM.  The reference has no FFI or
 * plugin interface; its seam is the set of free functions in kalmanFilter/modules/1PointRansacEKF/ headers that
 * EKF::step calls.  Each entry point below replaces one of them (file:line given per function, relative to
 * /root/reference/kalmanFilter/modules/, EKF/ = 1PointRansacEKF/).  The C++ headers under
 * openekfmonoslam_amd/compat/ re-create the reference's own signatures on top of this ABI.
 *
 * Conventions
 *  - plain pointers and sizes only, caller-owned buffers, int status codes (ekf_types.h), no exceptions;
 *  - one engine = one host thread + its own HIP stream; several engines may coexist (no globals);
 *  - state, covariance P, the map and all per-frame intermediates stay resident in HBM between calls;
 *    ekf_set_state / ekf_get_state are the explicit host<->device synchronisation points;
 *  - predictions, Jacobian blocks and the H*P row pairs produced by ekf_predict_measurements stay on the
 *    device, keyed by featureIndex; ekf_match / ekf_ransac / ekf_update / ekf_rescue consume them, which is how
 *    the reference's index-aligned (prediction, Jacobian, match) vectors (EKF/Update.h:47) are expressed here.
 *  - there is no CPU fallback: every compute entry point returns EKF_ERR_NO_DEVICE without a usable GPU.
 */
#ifndef EKF_ENGINE_H
#define EKF_ENGINE_H

#include <stddef.h>

#include "ekf_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct EkfEngine EkfEngine;

enum {
    EKF_PRECISION_F64 = 0, /* covariance and work matrices in fp64 (reference arithmetic)                   */
    EKF_PRECISION_F32 = 1, /* covariance, H*P and the rank-m downdate in fp32 (MFMA f32); S, its Cholesky     */
                           /* factor, the state and all Jacobians stay fp64                                   */
    EKF_PRECISION_F32_EXACT = 2, /* fp32 STORAGE of the covariance and of H*P, reference-class arithmetic: B = inv(L) H P in
                                 * fp64 and the rank-m downdate P - B'B accumulated EXACTLY (int8 digit planes of B on the
                                 * int8 MFMA, int32 sums), rounded to fp32 once per entry and update.  The reference computes
                                 * in double throughout (Core/Base.h:67; Update.cpp:105-108, 214-218). */
    EKF_PRECISION_F64_EXACT = 3, /* fp64 STORAGE of the covariance with the same exact int8 update: the sums are subtracted from an
                                 * fp64-stored P, no rounding to fp32 -- for maps on which fp32 storage alone leaves the reference's
                                 * 1e-5 (fresh maps of >= ~1400 features); about 2.5 x the speed of EKF_PRECISION_F64 there. */
    EKF_PRECISION_AUTO = 4      /* the fastest configuration that holds every feature parameter within 1e-5 of the fp64 reference
                                 * at this capacity (measured): EKF_PRECISION_F32_EXACT up to
                                 * EKF_AUTO_F32_MAX_FEATURES features, EKF_PRECISION_F64_EXACT up to EKF_AUTO_EXACT_MAX_FEATURES,
                                 * EKF_PRECISION_F64 above; ekf_get_precision reports the choice */
};
#define EKF_AUTO_F32_MAX_FEATURES 1024
#define EKF_AUTO_EXACT_MAX_FEATURES 2048

typedef struct EkfEngineConfig {
    EkfCamera cam;
    EkfParams par;
    int32_t max_features;  /* map capacity N_max; state capacity is 13 + 6 N_max                            */
    int32_t max_keypoints; /* per-frame keypoint capacity                                                    */
    int32_t precision;     /* EKF_PRECISION_*                                                                */
    int32_t device;        /* HIP device ordinal, or -1 for the current device                               */
    int32_t ransac_batch;  /* hypotheses evaluated per launch (0 = default 32)                               */
    int32_t flags;         /* descriptor format, EKF_DESCRIPTOR_* (0 = 32-byte binary / Hamming)                  */
} EkfEngineConfig;

/* Descriptor format of the map features and keypoints = the two branches of computeDistance
 * (EKF/Matching.cpp:47-92):
 *   EKF_DESCRIPTOR_U8_HAMMING      CV_8U, EKF_DESC_BYTES bytes per descriptor, Hamming distance (:76-92; BRIEF-32, the
 *                                  shipped configuration)
 *   EKF_DESCRIPTOR_F32_L2(cols)    CV_32F, `cols` floats per descriptor (1..1024), L2 distance (:60-73; SURF / SIFT
 *                                  style extractors of Cfg/DescriptorExtractorFactory.cpp)
 * Every `desc32` argument of this ABI is a row-major matrix with ekf_descriptor_bytes(e) bytes per row. */
#define EKF_DESCRIPTOR_U8_HAMMING 0
#define EKF_DESCRIPTOR_F32_L2(cols) (1 | ((cols) << 8))

/* Per-step counters; same layout as the oracle's OrcStepInfo. */
typedef struct EkfStepInfo {
    int32_t n_predicted;
    int32_t n_matches;
    int32_t n_hypotheses;
    int32_t n_inliers;
    int32_t n_outliers;
    int32_t n_rescued;
    int32_t status;
    int32_t n_sweep_retries; /* updates of this step whose persistent Cholesky sweep timed out and that were run again, transparently,
                              * on the launch-per-panel sweep (the slot the oracle's OrcStepInfo leaves as padding; 0 in normal
                              * operation; ekf_get_sweep_retries totals them) */
} EkfStepInfo;

/* Accumulated GPU time per stage, in milliseconds, under the reference's own stage names
 * (EKF/EKF.cpp:291,344,410,437,513,539) plus the dominant kernel on its own. */
typedef struct EkfStageTimes {
    double prediction_ms;
    double matching_ms;
    double ransac_ms;
    double update_li_ms;
    double rescue_ms;
    double update_hi_ms;
    double p_update_kernel_ms; /* sum over launches of the P <- P - B'B kernel (HIP events on the engine stream) */
    int64_t p_update_launches;
    double p_update_flops;     /* algorithmic flops of those launches: n^2 * m each (upper triangle)          */
    double p_update_bytes;     /* algorithmic bytes: 2 * n^2 * w / 2 ... see DESIGN.md                        */
    int64_t steps;
} EkfStageTimes;

/* -- life cycle ------------------------------------------------------------------------------------------ */
/* Replaces EKF::EKF (EKF/EKF.cpp:124-165) minus file I/O: the two config PODs are passed by value. */
int ekf_engine_create(const EkfEngineConfig *cfg, EkfEngine **out);
void ekf_engine_destroy(EkfEngine *e);
const char *ekf_last_error(const EkfEngine *e);
int ekf_abi_version(void);
/* Number of usable HIP devices (0 when none); never fails. */
int ekf_device_count(void);

/* -- host <-> device synchronisation --------------------------------------------------------------------- */
/* Loads State (EKF/State.h:40-81: position, orientation, linearVelocity, angularVelocity as x13 =
 * r(3) q(4: w x y z) v(3) w(3); mapFeatures as 6 doubles per feature + type + 32-byte descriptor) and
 * EKF::stateCovarianceMatrix (n x n row-major doubles, leading dimension n).  Covariance positions are assigned
 * in map order exactly as MapFeature::covarianceMatrixPos (EKF/MapFeature.h:68). */
int ekf_set_state(EkfEngine *e, const double x13[13], int n_features, const double *feature_pos,
                  const int32_t *feature_type, const uint8_t *desc32, const double *P);
/* Any output pointer may be NULL.  P is written as n x n doubles. */
int ekf_get_state(EkfEngine *e, double x13[13], double *feature_pos, double *P);
/* Per-feature bookkeeping updateMapFeatures maintains every step (EKF/MapManagement.cpp:77-113): the current map
 * descriptors (32 bytes each) and MapFeature::timesPredicted / timesMatched.  Any pointer may be NULL. */
int ekf_get_map_features(EkfEngine *e, uint8_t *desc32, uint32_t *times_predicted, uint32_t *times_matched);
int ekf_state_dim(const EkfEngine *e);
int ekf_descriptor_bytes(const EkfEngine *e); /* bytes per descriptor row (32, or 4 * cols for CV_32F) */
int ekf_get_precision(const EkfEngine *e);    /* the EKF_PRECISION_* in use (what EKF_PRECISION_AUTO resolved to) */
int ekf_num_features(const EkfEngine *e);

/* -- map management: the state dimension changes, P is edited in place on the device ------ */
/* initState + initCovariance                              EKF/CommonFunctions.cpp:39-80 (empties the map) */
int ekf_reset(EkfEngine *e);
/* addFeaturesToStateAndCovariance(measurements, state, P) EKF/AddMapFeature.h (.cpp:354; per feature :293-344,
 * covariance :221-289): uv = distorted pixel of each new feature (2 doubles each), desc32 = its descriptor. */
int ekf_add_features(EkfEngine *e, const double *uv, const uint8_t *desc32, int count);
/* removeFeaturesFromStateAndCovariance(featuresToRemove, state, P)  EKF/MapManagement.cpp:212-259;
 * ascending feature indices (map order, as the reference requires sorted ranges). */
int ekf_remove_features(EkfEngine *e, const int32_t *feat_idx, int count);
/* removeBadMapFeatures(state, P)                          EKF/MapManagement.cpp:279-308 */
int ekf_remove_bad_features(EkfEngine *e, int *n_removed);
/* convertMapFeaturesInverseDepthToDepth(state, P)         EKF/MapManagement.cpp:494-521 (at most one per call;
 * converted_index receives the feature index or -1) */
int ekf_convert_inverse_depth_to_depth(EkfEngine *e, int *converted_index);
/* the 13x13 camera block of the covariance, row-major: what EKF::step logs as StateCovarianceMatrixEstimation
 * (EKF/EKF.cpp:626) -- without moving the whole matrix */
int ekf_get_camera_covariance(EkfEngine *e, double P13[169]);
/* features the last full prediction did not see (unseenFeatures of predictCameraMeasurements,
 * EKF/MeasurementPrediction.cpp:705; EKF::step removes them under the conditions of EKF/EKF.cpp:583-592).
 * Ascending feature indices; feat_idx may be NULL to get the count only. */
int ekf_get_unseen_features(EkfEngine *e, int32_t *feat_idx, int *count);
/* MapFeature::featureType / covarianceMatrixPos of every feature (EKF/MapFeature.h:60-68) */
int ekf_get_feature_layout(EkfEngine *e, int32_t *type, int32_t *covpos);
/* predictedDistortedFeatures of the step's predictCameraMeasurements as they were BEFORE the updates -- what
 * drawPrediction receives for the per-frame prediction image (EKF/EKF.cpp:294-305, Gui/Draw.cpp:266-310).  Off by
 * default (one extra small launch per frame when on); preds may be NULL to get the count only. */
int ekf_keep_step_predictions(EkfEngine *e, int on);
int ekf_get_step_predictions(EkfEngine *e, EkfPrediction *preds, int *n_preds);

/* -- stages ---------------------------------------------------------------------------------------------- */
/* stateAndCovariancePrediction(State&, Matd&)            EKF/StateAndCovariancePrediction.h:41 (.cpp:244-253) */
int ekf_predict(EkfEngine *e);

/* predictCameraMeasurements(state, P, features, featureIndexes, preds, jacobians, notPredicted)
 *                                                        EKF/MeasurementPrediction.h:59 (.cpp:705-719)
 * feat_idx == NULL / count == 0 predicts every map feature.  Outputs (all optional): predictions in input order,
 * Hs (2x13 doubles each) and Hf (2x6 doubles each) Jacobian blocks. */
int ekf_predict_measurements(EkfEngine *e, const int32_t *feat_idx, int count, EkfPrediction *preds, int *n_preds,
                             double *Hs, double *Hf);

/* predictMeasurementState(state, features, featureIndexes, preds, notPredicted)
 *                                                        EKF/MeasurementPrediction.h:41 (.cpp:203-265)
 * on the engine's current state, all features; no Jacobians, device tables untouched. */
int ekf_predict_measurement_state(EkfEngine *e, EkfPrediction *preds, int *n_preds);

/* matchPredictedFeatures(image, features, predictions, matches)   EKF/Matching.h:66 (.cpp:181-264),
 * the stage downstream of the detector/descriptor (ellipse gate + descriptor distance + 2-best selection,
 * .cpp:217-262) for the keypoints of this frame, against the predictions of the last full
 * ekf_predict_measurements call. */
int ekf_match(EkfEngine *e, const EkfKeypoint *kps, const uint8_t *desc32, int n_kp, EkfMatch *matches,
              int *n_matches);

/* ransac(state, P, preds, jacobians, matches, inliers, inlierPreds, inlierJacobians, outliers)
 *                                                        EKF/1PointRansac.h:42 (.cpp:101-234)
 * inlier_mask[i] = 1 for matches kept as low-innovation inliers.  n_hypotheses (optional) = hypotheses the
 * sequential reference loop would have evaluated. */
int ekf_ransac(EkfEngine *e, const EkfMatch *matches, int M, uint8_t *inlier_mask, int *n_hypotheses);
/* ekf_ransac / ekf_update / ekf_update_only_state / ekf_rescue: at most ONE match per featureIndex (what
 * matchPredictedFeatures produces); a list with a repeated featureIndex returns EKF_ERR_INVALID_ARG. */

/* update(state, P, matches, preds, jacobians)            EKF/Update.h:48 (.cpp:282-319) */
int ekf_update(EkfEngine *e, const EkfMatch *matches, int M);

/* updateOnlyState(preds, matches, jacobians, state, P)   EKF/Update.h:42 (.cpp:269-275) */
int ekf_update_only_state(EkfEngine *e, const EkfMatch *matches, int M);

/* rescueOutliers(outlierMatches, preds, jacobians, rescued...)    EKF/EKF.cpp:68-119
 * against the predictions of the last ekf_predict_measurements call that covered those features. */
int ekf_rescue(EkfEngine *e, const EkfMatch *outliers, int M, uint8_t *rescued_mask);

/* EKF::step(image) with a fixed map                      EKF/EKF.h:57 (.cpp:242-572, including updateMapFeatures),
 * fed the frame's keypoints. */
int ekf_step(EkfEngine *e, const EkfKeypoint *kps, const uint8_t *desc32, int n_kp, EkfStepInfo *info);

/* -- pre-staged sequences (inputs resident in HBM before a timed region) ---------------------------------- */
int ekf_frames_upload(EkfEngine *e, int n_frames, const int32_t *kp_counts, const EkfKeypoint *kps_concat,
                      const uint8_t *desc_concat);
int ekf_step_frame(EkfEngine *e, int frame, EkfStepInfo *info);
/* on: ekf_step / ekf_step_frame return without the final read-back of the error flag (they then return as soon as the
 * second update is ENQUEUED).  A failed factorisation of that update (EKF_ERR_NOT_POSITIVE_DEFINITE) is reported by the
 * next ekf_step*, ekf_synchronize or ekf_get_state, whichever comes first (the flag is sticky on the device until one of
 * them reports and clears it) -- the reference reports nothing at all (cv::invert returns zeros).  Default off. */
int ekf_set_async_errors(EkfEngine *e, int on);
/* How an update forms B = inv(L) (H P), the factor of the covariance downdate (replaces K = P H' inv(S),
 * EKF/Update.cpp:105-108): EKF_UPDATE_PATH_AUTO by the number of measurement rows (the default), EKF_UPDATE_PATH_SWEEP
 * always inside the launches of the Cholesky sweep (forward substitution, no explicit inverse), EKF_UPDATE_PATH_GEMM
 * always by inverting L and one GEMM.  Same result to rounding; a tuning / test knob. */
enum { EKF_UPDATE_PATH_AUTO = 0, EKF_UPDATE_PATH_SWEEP = 1, EKF_UPDATE_PATH_GEMM = 2 };
int ekf_set_update_path(EkfEngine *e, int path);
/* How the blocked Cholesky sweep of S (replaces S.inv(), EKF/Update.cpp:108) is launched: EKF_SWEEP_SINGLE one 32-row panel per
 * launch, EKF_SWEEP_PAIRS two panels per launch with a 64 x 64 look-ahead inverse, EKF_SWEEP_PERSISTENT ONE launch per update
 * (a resident critical workgroup factorises panel after panel, tile and row-block workers follow it through flags; updates of at
 * most 2048 rows on an unsharded engine in the fp64 and the exact configuration whose grid fits the device, otherwise launches as
 * under AUTO; a sweep whose waits time out is run again on the launch-per-panel path, see ekf_get_sweep_retries),
 * EKF_SWEEP_LAUNCHES the round-4 rule (single launches while a launch is bound by its look-ahead factorisation, pairs once the
 * rows of B = inv(L) H P are the longest role), EKF_SWEEP_AUTO (the default): persistent where it applies AND the map has fewer
 * than 8192 state columns (above, its row-block workers take several blocks of columns each and two panels per launch are faster:
 * N = 2000, 14.0 against 20.4 us per panel), else LAUNCHES.  Same result to rounding; a tuning / test knob. */
enum { EKF_SWEEP_PAIRS = 0, EKF_SWEEP_SINGLE = 1, EKF_SWEEP_AUTO = 2, EKF_SWEEP_PERSISTENT = 3, EKF_SWEEP_LAUNCHES = 4 };
int ekf_set_sweep_mode(EkfEngine *e, int mode);

/* -- matcher mode B: image in, no detector ----------------------------------------------------------------
 * matchPredictedFeatures(const cv::Mat &image, ...)  EKF/Matching.h:66 and EKF::step(const cv::Mat &image)
 * EKF/EKF.h:57 take the frame itself.  Mode B keeps that signature: the frame is uploaded (1, 3 = BGR or
 * 4 = RGBA bytes per pixel, as Img/FileSequenceImageGenerator.cpp:82 and android jni/EKFNative.cpp:163 deliver
 * it), reduced to a 3-level gray pyramid on the device, and each predicted feature is matched by zero-mean NCC
 * of its 11x11 template inside the predicted ellipse (gate of Matching.cpp:217-241), coarse to fine.  The
 * reference has no such matcher (it runs an OpenCV detector + descriptor on the host, Matching.cpp:188-210);
 * the definition is this build's and is restated on the CPU in oracle/ekf_oracle.c for the parity tests.
 * Matches carry keypointIndex = -1 and integer pixel positions; templates are captured once and kept. */
int ekf_image_upload(EkfEngine *e, const uint8_t *image, int width, int height, int stride, int channels);
int ekf_get_image_level(EkfEngine *e, int level, uint8_t *out, int *width, int *height); /* readback for tests */
/* templates of the listed map features, cut from the CURRENT image around uv (the pixel a feature was
 * initialised at: addFeaturesToStateAndCovariance keeps the descriptor there, EKF/AddMapFeature.cpp:317-337) */
int ekf_capture_templates(EkfEngine *e, const int32_t *feat_idx, const double *uv, int count);
int ekf_match_ncc(EkfEngine *e, EkfMatch *matches, int *n_matches);
int ekf_step_image(EkfEngine *e, const uint8_t *image, int width, int height, int stride, int channels,
                   EkfStepInfo *info);
/* detectNewImageFeatures(image, featuresPrediction, newImageFeaturesMaxSize, newImageFeatures)
 *                                                              EKF/DetectNewImageFeatures.cpp:337
 * on the CURRENT image, masked by the gate ellipses of the last full prediction made while an image was loaded
 * (buildImageMask :102).  The detector is this build's own (integer Harris-type measure, best unmasked pixel per
 * 16x16 cell, kernels_detect.hip; the reference's OpenCV STAR is third-party code that is not here); the zone
 * heuristic is the reference's searchFeaturesByZone (:171) with rand() replaced by "strongest candidate of the
 * zone".  divide_times = DetectNewFeaturesImageAreasDivideTimes, mask_ellipse_size =
 * DetectNewFeaturesImageMaskEllipseSize, min_response = detector threshold on the integer measure.
 * Writes up to max_new pixel positions (x, y) to uv_out; the caller then adds them (ekf_add_features) and cuts their
 * templates (ekf_capture_templates). */
int ekf_detect_new_features(EkfEngine *e, int max_new, int divide_times, double mask_ellipse_size,
                            double min_response, double *uv_out, int *count);

/* pre-staged image sequence (n_frames images of identical geometry, concatenated) */
int ekf_images_upload(EkfEngine *e, int n_frames, const uint8_t *images, int width, int height, int stride,
                      int channels);
int ekf_select_staged_image(EkfEngine *e, int frame); /* build the pyramid of a staged frame */
int ekf_step_staged_image(EkfEngine *e, int frame, EkfStepInfo *info);

/* -- instrumentation ------------------------------------------------------------------------------------- */
int ekf_timing_enable(EkfEngine *e, int on); /* HIP-event timing of stages and of the P-update kernel */
int ekf_timing_reset(EkfEngine *e);
int ekf_timing_get(EkfEngine *e, EkfStageTimes *out);
int ekf_synchronize(EkfEngine *e);
/* Per-launch record of the P-update kernel since the last ekf_timing_reset: rows m of B (k-depth) and the
 * HIP-event duration in ms.  Writes at most `capacity` entries; *count receives the number available. */
int ekf_timing_p_update_launches(EkfEngine *e, int capacity, int32_t *m_rows, float *ms, int *count);
/* The blocked Cholesky sweep of S = H P H' + R (replaces S.inv(), EKF/Update.cpp:108) since the last ekf_timing_reset:
 * HIP-event time over the sweep's launches of every update, their number (panels of 32 rows) and of updates, the fp64
 * flops of the factorisation (m^3 / 3 each) and the flops of B = inv(L) (H P) formed in the same launches (m^2 n each;
 * 0 for updates that went through the explicit inverse + GEMM).  Any output pointer may be null. */
int ekf_timing_sweep(EkfEngine *e, double *kernel_ms, int64_t *panels, int64_t *updates, double *flops_fp64, double *flops_b);
/* ... and the number of kernel launches behind those panels (a launch of the two-panels-per-launch scheme covers two), plus,
 * EKF_PRECISION_F32_EXACT only, the HIP-event time between the end of each update's sweep and the start of its downdate kernel
 * (dx, the state update and, above 2048 rows, the triangular inverse and the int8 GEMM B = inv(L) G). */
int ekf_timing_sweep_launches(EkfEngine *e, int64_t *launches, double *slice_ms);
/* Measurement aid, EKF_PRECISION_F64 engines only: replaces every entry of the device-resident covariance by its nearest fp32
 * value, in place (asynchronous, on the engine's stream).  Called between the stage functions at the points where an
 * fp32-storage configuration rounds P (after ekf_set_state, ekf_predict and each ekf_update) it yields the filter "fp32 storage,
 * fp64 arithmetic": the accuracy floor of EKF_PRECISION_F32 / _F32_EXACT on a sequence (scripts/storage_floor_gpu.py).  The
 * reference keeps everything in double (Core/Base.h:67). */
int ekf_round_covariance_to_f32(EkfEngine *e);
/* Updates re-run on the launch-per-panel sweep since the engine was created because their persistent sweep
 * (EKF_SWEEP_PERSISTENT / _AUTO) timed out: the sweep needs all of its workgroups resident at once, which another process's
 * kernels on the device or a device partition can prevent; its waits are bounded (30 ms), the kernels behind a failed sweep leave
 * x and P untouched, and the engine runs the same update again from the same P -- the caller sees EKF_OK and this counter.
 * (The fault-injection hook that exercises this path is declared in ekf_test_hooks.h, not here.) */
int ekf_get_sweep_retries(const EkfEngine *e);

/* -- row-sharded filter (multi-GPU) ----------------------------------------------------------
 * One engine per GPU / rank.  Rank g stores the 13 camera rows of P (replicated, updated identically everywhere)
 * and the rows of the features [f0_g, f1_g) it owns, each with all n columns; everything else (state, map, H.P rows,
 * S, its factor, B) is replicated.  Owned rows are predicted, multiplied (H_f P) and downdated locally; the one
 * exchange per prediction is an all-gather of the H.P row blocks (and of the 2x2 S_i of the owned features), done by
 * the callback the host installs: RCCL broadcasts/all-gather between processes, device-to-device copies when
 * several ranks share a GPU (tests).  Map management is not available on a sharded engine.
 * EKF_PRECISION_F32_EXACT (round 4): B is NOT replicated -- every rank forms the columns of B = inv(L) H P that belong to the state
 * rows it holds and the ranks all-gather B's int8 digit planes (EKF_XCHG_BPLANES; EKF_XCHG_PDIAG carries the diagonal of P behind
 * their column scales); for updates of at most 2048 rows no rows of H P travel either: G[:, own columns] follows from the own rows
 * of P by symmetry and S is assembled by block columns and all-gathered (EKF_XCHG_SCOLS).  DESIGN.md section 8. */
typedef int (*EkfExchangeFn)(void *user, int what, void *device_base, size_t row_bytes, const int32_t *row_begin,
                             int world, int rank);
enum { EKF_XCHG_HP = 0, EKF_XCHG_PRED_S = 1, EKF_XCHG_HPC = 2, EKF_XCHG_PDIAG = 3, EKF_XCHG_BPLANES = 4, EKF_XCHG_SCOLS = 5,
       /* round 6 -- matching and RANSAC divided by feature ownership: the per-prediction match tables (valid flag,
        * keypoint index or matched pixel, distance; rows = prediction slots) and a RANSAC batch's support counts and inlier masks
        * (rows = hypotheses of the batch) */
       EKF_XCHG_MATCH_VALID = 6, EKF_XCHG_MATCH_KP = 7, EKF_XCHG_MATCH_DIST = 8, EKF_XCHG_MATCH_XY = 9,
       EKF_XCHG_HYP_COUNT = 10, EKF_XCHG_HYP_FLAGS = 11 };
/* what: which replicated table is being completed; device_base: its first row on this rank's GPU; rank r owns rows
 * [row_begin[r], row_begin[r+1]) of row_bytes each and has just written them.  The callback returns when this
 * rank's table holds every rank's rows (0 = ok).  It is called with the engine's stream idle. */
int ekf_engine_create_sharded(const EkfEngineConfig *cfg, int rank, int world, EkfEngine **out);
int ekf_set_exchange(EkfEngine *e, EkfExchangeFn fn, void *user);
/* In-engine transport, one process per GPU: an RCCL communicator owned by the engine.  The exchange is then a grouped
 * ncclSend / ncclRecv of the row blocks between every pair of ranks (the direct schedule for point-to-point xGMI: every
 * rank's block travels over its own link to each peer), enqueued on the engine's stream -- no stream drain, no host
 * callback.  Rank 0 obtains the id, the host distributes it (any out-of-band channel), every rank calls
 * ekf_comm_init (collective).  RCCL is loaded at run time (librccl.so.1); EKF_ERR_COMM if it is not there.
 * A callback installed with ekf_set_exchange takes precedence over the communicator (a host whose ranks could not all
 * form one falls back to its own transport on every rank); ekf_set_exchange(e, NULL, NULL) hands it back. */
#define EKF_COMM_ID_BYTES 128
int ekf_comm_unique_id(uint8_t id[EKF_COMM_ID_BYTES]);
int ekf_comm_init(EkfEngine *e, const uint8_t id[EKF_COMM_ID_BYTES]);
/* P rows held by this rank: [*row_begin, *row_end) of the state (rank 0's range starts at 0: the camera rows,
 * which every rank also keeps).  On a sharded engine ekf_set_state takes the full n x n matrix and keeps its rows;
 * ekf_get_state fills only those rows of the caller's n x n buffer (camera rows + owned rows). */
int ekf_shard_info(const EkfEngine *e, int *rank, int *world, int *row_begin, int *row_end);
/* EKF_PRECISION_F32_EXACT on a sharded engine: a rank forms the rows of B = inv(L) H P for its own column blocks only and receives
 * the others' int8 digit planes (5 bytes per element of B).  plane_bytes_received: total since creation; [own_columns_begin,
 * own_columns_end): the columns whose rows of B this rank formed in its LAST update -- they end exactly at the ownership boundaries
 * on the by-symmetry route (updates of <= 2048 rows) and at multiples of 32 on the inverse + GEMM route; before the first update:
 * the ownership boundaries rounded to 32.  Tests compare both with the cost model. */
int ekf_shard_counters(EkfEngine *e, int64_t *plane_bytes_received, int32_t *own_columns_begin, int32_t *own_columns_end);
/* device-to-device copy on the engine's device, synchronous: the exchange primitive when ranks share a GPU */
int ekf_device_copy(EkfEngine *e, void *dst, const void *src, size_t bytes);

/* -- partition of the covariance rows across ranks; pure host arithmetic ------------------------------------- */
/* Row block [row_begin, row_end) of P owned by `rank` of `world` for a map of n_features inverse-depth features:
 * the 13 camera rows go to rank 0, features are split contiguously. */
int ekf_shard_rows(int n_features, int world, int rank, int *row_begin, int *row_end);

#ifdef __cplusplus
}
#endif
#endif /* EKF_ENGINE_H */
