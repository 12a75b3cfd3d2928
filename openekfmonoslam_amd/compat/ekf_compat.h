// ekf_compat.h -- the reference's own C++ surface for the hot path, re-created on top of the C ABI
// (include/ekf_engine.h).  Header-only, C++11, no OpenCV, no HIP headers: it links against libekf_engine.so only.
//
// What is mirrored (names, argument order and meaning; citations relative to
// /root/reference/kalmanFilter/modules/1PointRansacEKF/):
//   types      State (State.h:40-81), MapFeature (MapFeature.h:48-78), ImageFeaturePrediction
//              (ImageFeaturePrediction.h:37-50), FeatureMatch (Matching.h:38-48), Matd / VectorMatd (Core/Base.h:171-174;
//              here a minimal owning row-major matrix, not cv::Mat_)
//   functions  stateAndCovariancePrediction (StateAndCovariancePrediction.h:41), predictMeasurementState
//              (MeasurementPrediction.h:41), predictCameraMeasurements (MeasurementPrediction.h:59),
//              matchPredictedFeatures (Matching.h:66; takes the frame's keypoints + descriptors instead of the image:
//              the detector/descriptor stage is outside the hot path), ransac (1PointRansac.h:42), updateOnlyState
//              (Update.h:42), update (Update.h:48), rescueOutliers (EKF.cpp:68)
//   class      EKF { init, step, state, stateCovarianceMatrix } (EKF.h:41-63) -- device-resident between steps.
//
// Error behaviour: the reference signals nothing (void everywhere).  Here a failing engine call throws
// ekf_compat::Error carrying the C status code; a singular S is reported instead of silently producing zeros.
//
// The free functions are stateless like the reference's: each one uploads (State, P), runs its stage on the GPU and
// downloads the results -- correct but PCIe-bound.  Jacobian vectors handed to ransac/update/updateOnlyState must
// be the ones predictCameraMeasurements returned for the same (state, P) (the reference's own contract,
// Update.h:47); the wrappers recompute them on the device rather than uploading dense 2 x n matrices.
// The EKF class is the efficient path: everything stays in HBM across init()/step().
#ifndef EKF_COMPAT_H
#define EKF_COMPAT_H

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ekf_engine.h"

// ---------------------------------------------------------------------------------------------------- matrices
class Matd {
public:
    int rows, cols;
    Matd() : rows(0), cols(0) {}
    Matd(int r, int c) : rows(r), cols(c), d_((size_t)r * c, 0.0) {}
    static Matd zeros(int r, int c) { return Matd(r, c); }
    static Matd eye(int r, int c)
    {
        Matd m(r, c);
        for (int i = 0; i < r && i < c; ++i) m[i][i] = 1.0;
        return m;
    }
    double *operator[](int i) { return d_.data() + (size_t)i * cols; }
    const double *operator[](int i) const { return d_.data() + (size_t)i * cols; }
    double *ptr() { return d_.data(); }
    const double *ptr() const { return d_.data(); }
    double &at(int i, int j) { return d_[(size_t)i * cols + j]; }
    Matd clone() const { return *this; }

private:
    std::vector<double> d_;
};
typedef std::vector<Matd *> VectorMatd;

// ------------------------------------------------------------------------------------------------ data types
enum MapFeatureType { MAPFEATURE_TYPE_INVALID, MAPFEATURE_TYPE_DEPTH, MAPFEATURE_TYPE_INVERSE_DEPTH };

struct Descriptor32 {
    uint8_t bytes[EKF_DESC_BYTES];
    Descriptor32() { std::memset(bytes, 0, sizeof(bytes)); }
};

class MapFeature {
public:
    MapFeature() : featureType(MAPFEATURE_TYPE_INVALID), positionDimension(0), covarianceMatrixPos(0), timesPredicted(0), timesMatched(0)
    {
        std::memset(pos_, 0, sizeof(pos_));
        position = pos_;
    }
    MapFeature(const double *p, int dim, int covPos, const Descriptor32 &desc, MapFeatureType type)
        : featureType(type), positionDimension(dim), covarianceMatrixPos(covPos), descriptor(desc), timesPredicted(0), timesMatched(0)
    {
        std::memset(pos_, 0, sizeof(pos_));
        std::memcpy(pos_, p, sizeof(double) * dim);
        position = pos_;
    }
    MapFeature(const MapFeature &o) { *this = o; }
    MapFeature &operator=(const MapFeature &o)
    {
        featureType = o.featureType;
        positionDimension = o.positionDimension;
        covarianceMatrixPos = o.covarianceMatrixPos;
        descriptor = o.descriptor;
        timesPredicted = o.timesPredicted;
        timesMatched = o.timesMatched;
        std::memcpy(pos_, o.pos_, sizeof(pos_));
        position = pos_;
        return *this;
    }
    MapFeatureType featureType;
    double *position;
    int positionDimension;
    int covarianceMatrixPos;
    Descriptor32 descriptor;
    unsigned timesPredicted, timesMatched;

private:
    double pos_[6];
};
typedef std::vector<MapFeature *> VectorMapFeature;

class State {
public:
    State() { init(); }
    State(const State &o)
    {
        init();
        std::memcpy(x_, o.x_, sizeof(x_));
        std::memcpy(R_, o.R_, sizeof(R_));
        for (size_t i = 0; i < o.mapFeatures.size(); ++i) addFeature(new MapFeature(*o.mapFeatures[i]));
    }
    ~State() { removeAllFeatures(); }
    void setOrientation(const double *q)
    { // State.cpp:131-139: copies q, recomputes R, never normalises
        for (int i = 0; i < 4; ++i) orientation[i] = q[i];
        const double r = q[0], x = q[1], y = q[2], z = q[3];
        double *M = orientationRotationMatrix;
        M[0] = r * r + x * x - y * y - z * z; M[1] = 2 * (x * y - r * z); M[2] = 2 * (z * x + r * y);
        M[3] = 2 * (x * y + r * z); M[4] = r * r - x * x + y * y - z * z; M[5] = 2 * (y * z - r * x);
        M[6] = 2 * (z * x - r * y); M[7] = 2 * (y * z + r * x); M[8] = r * r - x * x - y * y + z * z;
    }
    void addFeature(MapFeature *f)
    {
        mapFeatures.push_back(f);
        if (f->featureType == MAPFEATURE_TYPE_DEPTH) mapFeaturesDepth.push_back(f);
        else mapFeaturesInvDepth.push_back(f);
    }
    void removeAllFeatures()
    {
        for (size_t i = 0; i < mapFeatures.size(); ++i) delete mapFeatures[i];
        mapFeatures.clear();
        mapFeaturesDepth.clear();
        mapFeaturesInvDepth.clear();
    }
    double *position, *orientation, *orientationRotationMatrix, *linearVelocity, *angularVelocity;
    VectorMapFeature mapFeaturesDepth, mapFeaturesInvDepth, mapFeatures;
    // packed r q v w, as the C ABI wants it
    const double *x13() const { return x_; }
    double *x13() { return x_; }

private:
    State &operator=(const State &);
    void init()
    {
        std::memset(x_, 0, sizeof(x_));
        position = x_;
        orientation = x_ + 3;
        linearVelocity = x_ + 7;
        angularVelocity = x_ + 10;
        orientationRotationMatrix = R_;
        const double q[4] = {1, 0, 0, 0};
        setOrientation(q);
    }
    double x_[13], R_[9];
};

class ImageFeaturePrediction {
public:
    ImageFeaturePrediction() : featureIndex(-1), covarianceMatrix(Matd::zeros(2, 2)) { imagePos[0] = imagePos[1] = -1.0; }
    ImageFeaturePrediction(int idx, const double *pos) : featureIndex(idx), covarianceMatrix(Matd::zeros(2, 2))
    {
        imagePos[0] = pos[0];
        imagePos[1] = pos[1];
    }
    int featureIndex;
    double imagePos[2];
    Matd covarianceMatrix;
};
typedef std::vector<ImageFeaturePrediction *> VectorImageFeaturePrediction;

class FeatureMatch {
public:
    FeatureMatch() : featureIndex(-1), distance(-1) { imagePos[0] = imagePos[1] = 0; }
    int featureIndex;
    double imagePos[2];
    Descriptor32 imagePosDescriptor;
    float distance;
};
typedef std::vector<FeatureMatch *> VectorFeatureMatch;

// the output of the (out-of-scope) detector + descriptor extractor for one frame
struct FrameKeypoints {
    std::vector<EkfKeypoint> keypoints;
    std::vector<uint8_t> descriptors; // 32 bytes per keypoint
};

// --------------------------------------------------------------------------------------------------- plumbing
namespace ekf_compat {

class Error : public std::runtime_error {
public:
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
    int code;
};

// Replaces the ConfigurationManager singleton (Configuration/ConfigurationManager.h:45-64) for the two PODs the
// hot path reads.  One process-wide engine serves the stateless free functions.
class Context {
public:
    static Context &instance()
    {
        static Context c;
        return c;
    }
    void configure(const EkfCamera &cam, const EkfParams &par, int maxFeatures, int precision = EKF_PRECISION_F64)
    {
        release();
        cfg_ = EkfEngineConfig();
        cfg_.cam = cam;
        cfg_.par = par;
        cfg_.max_features = maxFeatures;
        cfg_.max_keypoints = 0;
        cfg_.precision = precision;
        cfg_.device = -1;
        check(ekf_engine_create(&cfg_, &e_), "ekf_engine_create");
    }
    EkfEngine *engine()
    {
        if (!e_) throw Error(EKF_ERR_INVALID_ARG, "ekf_compat::Context not configured");
        return e_;
    }
    const EkfEngineConfig &config() const { return cfg_; }
    void check(int rc, const char *what)
    {
        if (rc != EKF_OK) throw Error(rc, std::string(what) + ": " + (e_ ? ekf_last_error(e_) : "engine creation failed"));
    }
    void release()
    {
        if (e_) ekf_engine_destroy(e_);
        e_ = 0;
    }
    ~Context() { release(); }

private:
    Context() : e_(0) {}
    EkfEngine *e_;
    EkfEngineConfig cfg_;
};

inline void chk(EkfEngine *e, int rc, const char *what)
{
    if (rc != EKF_OK) throw Error(rc, std::string(what) + ": " + (e ? ekf_last_error(e) : ""));
}

inline void upload(EkfEngine *e, const State &s, const Matd *P)
{
    const size_t N = s.mapFeatures.size();
    std::vector<double> pos(6 * N + 6, 0.0);
    std::vector<int32_t> type(N + 1);
    std::vector<uint8_t> desc(EKF_DESC_BYTES * N + EKF_DESC_BYTES);
    for (size_t i = 0; i < N; ++i) {
        const MapFeature *f = s.mapFeatures[i];
        std::memcpy(&pos[6 * i], f->position, sizeof(double) * f->positionDimension);
        type[i] = f->featureType == MAPFEATURE_TYPE_DEPTH ? EKF_FEATURE_DEPTH : EKF_FEATURE_INVERSE_DEPTH;
        std::memcpy(&desc[EKF_DESC_BYTES * i], f->descriptor.bytes, EKF_DESC_BYTES);
    }
    chk(e, ekf_set_state(e, s.x13(), (int)N, pos.data(), type.data(), desc.data(), P ? P->ptr() : 0), "ekf_set_state");
}

inline void download(EkfEngine *e, State &s, Matd *P)
{
    const size_t N = s.mapFeatures.size();
    std::vector<double> pos(6 * N + 6);
    double x[13];
    if (P && (P->rows != ekf_state_dim(e) || P->cols != P->rows)) *P = Matd(ekf_state_dim(e), ekf_state_dim(e));
    chk(e, ekf_get_state(e, x, pos.data(), P ? P->ptr() : 0), "ekf_get_state");
    std::memcpy(s.x13(), x, sizeof(x));
    s.setOrientation(x + 3);
    for (size_t i = 0; i < N; ++i) std::memcpy(s.mapFeatures[i]->position, &pos[6 * i], sizeof(double) * s.mapFeatures[i]->positionDimension);
}

inline std::vector<EkfMatch> packMatches(const VectorFeatureMatch &m)
{
    std::vector<EkfMatch> out(m.size());
    for (size_t i = 0; i < m.size(); ++i) {
        out[i].featureIndex = m[i]->featureIndex;
        out[i].keypointIndex = -1;
        out[i].imagePos[0] = m[i]->imagePos[0];
        out[i].imagePos[1] = m[i]->imagePos[1];
        out[i].distance = m[i]->distance;
        out[i]._pad = 0.f;
    }
    return out;
}

// device-side tables for exactly the features the caller's predictions refer to
inline void refreshTables(EkfEngine *e, const VectorImageFeaturePrediction &preds)
{
    std::vector<int32_t> idx(preds.size());
    for (size_t i = 0; i < preds.size(); ++i) idx[i] = preds[i]->featureIndex;
    int n = 0;
    if (!idx.empty()) Context::instance().check(ekf_predict_measurements(e, idx.data(), (int)idx.size(), 0, &n, 0, 0), "ekf_predict_measurements");
}

inline void emitPredictions(const State &state, const std::vector<EkfPrediction> &p, const std::vector<double> &Hs,
                            const std::vector<double> &Hf, int n, VectorImageFeaturePrediction &preds, VectorMatd *jac)
{
    int dim = 13;
    for (size_t i = 0; i < state.mapFeatures.size(); ++i) dim += state.mapFeatures[i]->positionDimension;
    for (int k = 0; k < n; ++k) {
        ImageFeaturePrediction *ip = new ImageFeaturePrediction(p[k].featureIndex, p[k].imagePos);
        for (int i = 0; i < 4; ++i) ip->covarianceMatrix.ptr()[i] = p[k].covarianceMatrix[i];
        preds.push_back(ip);
        if (jac) {
            const MapFeature *f = state.mapFeatures[p[k].featureIndex];
            Matd *J = new Matd(Matd::zeros(2, dim)); // dense 2 x n like MeasurementPrediction.cpp:688
            for (int r = 0; r < 2; ++r) {
                for (int c = 0; c < 13; ++c) (*J)[r][c] = Hs[(size_t)26 * k + r * 13 + c];
                for (int c = 0; c < f->positionDimension; ++c) (*J)[r][f->covarianceMatrixPos + c] = Hf[(size_t)12 * k + r * 6 + c];
            }
            jac->push_back(J);
        }
    }
}

} // namespace ekf_compat

// ------------------------------------------------------------------------------------- the reference's functions
inline void stateAndCovariancePrediction(State &state, Matd &covarianceMatrix)
{
    EkfEngine *e = ekf_compat::Context::instance().engine();
    ekf_compat::upload(e, state, &covarianceMatrix);
    ekf_compat::Context::instance().check(ekf_predict(e), "ekf_predict");
    ekf_compat::download(e, state, &covarianceMatrix);
}

inline void predictCameraMeasurements(const State &state, const Matd &predictedStateCovariance, const VectorMapFeature &features,
                                      const std::vector<int> &featureIndexes, VectorImageFeaturePrediction &predictedDistortedFeatures,
                                      VectorMatd &predictedFeatureJacobians, VectorMapFeature &notPredictedFeatures)
{
    EkfEngine *e = ekf_compat::Context::instance().engine();
    ekf_compat::upload(e, state, &predictedStateCovariance);
    const int cnt = (int)features.size();
    if (cnt == 0) return;
    std::vector<EkfPrediction> p(cnt);
    std::vector<double> Hs((size_t)26 * cnt), Hf((size_t)12 * cnt);
    std::vector<int32_t> idx(featureIndexes.begin(), featureIndexes.end());
    int n = 0;
    ekf_compat::Context::instance().check(
        ekf_predict_measurements(e, idx.empty() ? 0 : idx.data(), (int)idx.size(), p.data(), &n, Hs.data(), Hf.data()),
        "ekf_predict_measurements");
    ekf_compat::emitPredictions(state, p, Hs, Hf, n, predictedDistortedFeatures, &predictedFeatureJacobians);
    // features that were not predicted, in input order (MeasurementPrediction.cpp:260-263)
    int k = 0;
    for (int i = 0; i < cnt; ++i) {
        const int fi = idx.empty() ? i : idx[i];
        if (k < n && p[k].featureIndex == fi) ++k;
        else notPredictedFeatures.push_back(features[i]);
    }
}

inline void predictMeasurementState(const State &state, const VectorMapFeature &features, const std::vector<int> &featureIndexes,
                                    VectorImageFeaturePrediction &predictedFeatures, VectorMapFeature &notPredictedFeatures)
{
    EkfEngine *e = ekf_compat::Context::instance().engine();
    ekf_compat::upload(e, state, 0);
    const int N = (int)state.mapFeatures.size();
    if (features.empty()) return;
    std::vector<EkfPrediction> p(N + 1);
    int n = 0;
    ekf_compat::Context::instance().check(ekf_predict_measurement_state(e, p.data(), &n), "ekf_predict_measurement_state");
    std::vector<int> slot(N, -1);
    for (int k = 0; k < n; ++k) slot[p[k].featureIndex] = k;
    for (size_t i = 0; i < features.size(); ++i) {
        const int fi = featureIndexes.empty() ? (int)i : featureIndexes[i];
        if (slot[fi] >= 0) predictedFeatures.push_back(new ImageFeaturePrediction(fi, p[slot[fi]].imagePos));
        else notPredictedFeatures.push_back(features[i]);
    }
}

inline void matchPredictedFeatures(const FrameKeypoints &frame, const State &state, const Matd &covariance,
                                   const VectorImageFeaturePrediction &vectorfeaturePrediction, VectorFeatureMatch &matches)
{
    // the reference's signature is (image, features, predictions, matches): the image is replaced by the frame's
    // keypoints/descriptors, and (state, P) are passed because this stateless wrapper must rebuild the device tables
    EkfEngine *e = ekf_compat::Context::instance().engine();
    ekf_compat::upload(e, state, &covariance);
    int n = 0;
    ekf_compat::Context::instance().check(ekf_predict_measurements(e, 0, 0, 0, &n, 0, 0), "ekf_predict_measurements");
    (void)vectorfeaturePrediction;
    std::vector<EkfMatch> m(state.mapFeatures.size() + 1);
    int M = 0;
    ekf_compat::Context::instance().check(
        ekf_match(e, frame.keypoints.data(), frame.descriptors.data(), (int)frame.keypoints.size(), m.data(), &M), "ekf_match");
    for (int i = 0; i < M; ++i) {
        FeatureMatch *fm = new FeatureMatch;
        fm->featureIndex = m[i].featureIndex;
        fm->imagePos[0] = m[i].imagePos[0];
        fm->imagePos[1] = m[i].imagePos[1];
        fm->distance = m[i].distance;
        std::memcpy(fm->imagePosDescriptor.bytes, &frame.descriptors[(size_t)EKF_DESC_BYTES * m[i].keypointIndex], EKF_DESC_BYTES);
        matches.push_back(fm);
    }
}

inline void ransac(const State &state, Matd &covariance, const VectorImageFeaturePrediction &predictedMatchedFeatures,
                   const VectorMatd &predictedMatchedJacobians, const VectorFeatureMatch &matches, VectorFeatureMatch &inlierMatches,
                   VectorImageFeaturePrediction &inlierPredictions, VectorMatd &inlierJacobians, VectorFeatureMatch &outlierMatches)
{
    if (matches.empty()) return;
    EkfEngine *e = ekf_compat::Context::instance().engine();
    ekf_compat::upload(e, state, &covariance);
    ekf_compat::refreshTables(e, predictedMatchedFeatures);
    std::vector<EkfMatch> m = ekf_compat::packMatches(matches);
    std::vector<uint8_t> mask(m.size());
    ekf_compat::Context::instance().check(ekf_ransac(e, m.data(), (int)m.size(), mask.data(), 0), "ekf_ransac");
    inlierMatches.clear(); inlierPredictions.clear(); inlierJacobians.clear(); outlierMatches.clear();
    for (size_t i = 0; i < m.size(); ++i) {
        if (mask[i]) {
            inlierMatches.push_back(matches[i]);
            inlierPredictions.push_back(predictedMatchedFeatures[i]);
            inlierJacobians.push_back(predictedMatchedJacobians[i]);
        } else {
            outlierMatches.push_back(matches[i]);
        }
    }
}

inline void update(State &state, Matd &covariance, const VectorFeatureMatch &measurementMatchedFeatures,
                   const VectorImageFeaturePrediction &predictedDistortedFeatures, const VectorMatd &predictedFeatureJacobians)
{
    (void)predictedFeatureJacobians;
    if (measurementMatchedFeatures.empty()) return; // Update.cpp:292
    EkfEngine *e = ekf_compat::Context::instance().engine();
    ekf_compat::upload(e, state, &covariance);
    ekf_compat::refreshTables(e, predictedDistortedFeatures);
    std::vector<EkfMatch> m = ekf_compat::packMatches(measurementMatchedFeatures);
    ekf_compat::Context::instance().check(ekf_update(e, m.data(), (int)m.size()), "ekf_update");
    ekf_compat::download(e, state, &covariance);
}

inline void updateOnlyState(const VectorImageFeaturePrediction &predictedDistortedFeatures, const VectorFeatureMatch &measurementMatchedFeatures,
                            const VectorMatd &predictedFeatureJacobians, State &state, Matd &covariance)
{
    (void)predictedFeatureJacobians;
    if (measurementMatchedFeatures.empty()) return;
    EkfEngine *e = ekf_compat::Context::instance().engine();
    ekf_compat::upload(e, state, &covariance);
    ekf_compat::refreshTables(e, predictedDistortedFeatures);
    std::vector<EkfMatch> m = ekf_compat::packMatches(measurementMatchedFeatures);
    ekf_compat::Context::instance().check(ekf_update_only_state(e, m.data(), (int)m.size()), "ekf_update_only_state");
    ekf_compat::download(e, state, 0);
}

inline void rescueOutliers(const VectorFeatureMatch &outlierMatches, const VectorImageFeaturePrediction &outlierMatchFeaturePrediction,
                           const VectorMatd &outlierMatchFeaturePredictionJacobians, VectorFeatureMatch &rescuedMatches,
                           VectorImageFeaturePrediction &rescuedPredictions, VectorMatd &rescuedJacobians)
{
    rescuedMatches.clear(); rescuedPredictions.clear(); rescuedJacobians.clear();
    if (outlierMatches.empty()) return;
    // the 2x2 covariances travel inside the predictions (EKF.cpp:94), so this stage needs no device tables:
    // it is evaluated by the engine against the tables of the last predictCameraMeasurements call
    EkfEngine *e = ekf_compat::Context::instance().engine();
    std::vector<EkfMatch> m = ekf_compat::packMatches(outlierMatches);
    std::vector<uint8_t> mask(m.size());
    ekf_compat::Context::instance().check(ekf_rescue(e, m.data(), (int)m.size(), mask.data()), "ekf_rescue");
    for (size_t i = 0; i < m.size(); ++i)
        if (mask[i]) {
            rescuedMatches.push_back(outlierMatches[i]);
            rescuedPredictions.push_back(outlierMatchFeaturePrediction[i]);
            rescuedJacobians.push_back(outlierMatchFeaturePredictionJacobians[i]);
        }
}

// ------------------------------------------------------------------------------------------------ class EKF
// EKF.h:41-63.  The configuration file / output path of the reference constructor are replaced by the two PODs;
// init() takes an already seeded State + covariance (map management -- detecting and adding features -- is
// outside the hot path); step() takes the frame's keypoints.  state / stateCovarianceMatrix are refreshed from the
// device only when syncToHost() is called.
class EKF {
public:
    EKF(const EkfCamera &cam, const EkfParams &par, int maxFeatures, int precision = EKF_PRECISION_F64) : e_(0), steps_(0)
    {
        EkfEngineConfig cfg = EkfEngineConfig();
        cfg.cam = cam; cfg.par = par; cfg.max_features = maxFeatures; cfg.precision = precision; cfg.device = -1;
        const int rc = ekf_engine_create(&cfg, &e_);
        if (rc != EKF_OK) throw ekf_compat::Error(rc, "ekf_engine_create failed (no MI355X?)");
    }
    ~EKF() { if (e_) ekf_engine_destroy(e_); }
    void init(const State &seed, const Matd &P)
    {
        state.removeAllFeatures();
        std::memcpy(state.x13(), seed.x13(), 13 * sizeof(double));
        state.setOrientation(seed.orientation);
        for (size_t i = 0; i < seed.mapFeatures.size(); ++i) state.addFeature(new MapFeature(*seed.mapFeatures[i]));
        stateCovarianceMatrix = P;
        ekf_compat::upload(e_, state, &stateCovarianceMatrix);
    }
    EkfStepInfo step(const FrameKeypoints &frame)
    {
        EkfStepInfo info;
        const int rc = ekf_step(e_, frame.keypoints.data(), frame.descriptors.data(), (int)frame.keypoints.size(), &info);
        ++steps_;
        if (rc != EKF_OK) throw ekf_compat::Error(rc, ekf_last_error(e_));
        return info;
    }
    void syncToHost()
    {
        refreshLayout();
        ekf_compat::download(e_, state, &stateCovarianceMatrix);
    }
    // ---- map management on the device (EKF.cpp:574-612).  Detection of new features stays on the
    // host: the caller passes the distorted pixel + descriptor of each new ImageFeatureMeasurement.
    void addFeaturesToStateAndCovariance(const std::vector<double> &uv, const std::vector<uint8_t> &descriptors)
    { // AddMapFeature.cpp:354
        ekf_compat::chk(e_, ekf_add_features(e_, uv.data(), descriptors.empty() ? 0 : descriptors.data(), (int)(uv.size() / 2)),
                        "ekf_add_features");
    }
    int removeBadMapFeatures()
    { // MapManagement.cpp:279
        int k = 0;
        ekf_compat::chk(e_, ekf_remove_bad_features(e_, &k), "ekf_remove_bad_features");
        return k;
    }
    void removeFeaturesFromStateAndCovariance(const std::vector<int32_t> &ascendingFeatureIndexes)
    { // MapManagement.cpp:212
        ekf_compat::chk(e_, ekf_remove_features(e_, ascendingFeatureIndexes.data(), (int)ascendingFeatureIndexes.size()),
                        "ekf_remove_features");
    }
    int convertMapFeaturesInverseDepthToDepth()
    { // MapManagement.cpp:494
        int k = -1;
        ekf_compat::chk(e_, ekf_convert_inverse_depth_to_depth(e_, &k), "ekf_convert_inverse_depth_to_depth");
        return k;
    }
    EkfEngine *engine() { return e_; }
    Matd stateCovarianceMatrix;
    State state;

private:
    EKF(const EKF &);
    EKF &operator=(const EKF &);
    // rebuild the host-side MapFeature list when the device map changed size or parametrisation
    void refreshLayout()
    {
        const int N = ekf_num_features(e_);
        std::vector<int32_t> type(N + 1), covpos(N + 1);
        std::vector<uint8_t> desc((size_t)(N + 1) * EKF_DESC_BYTES);
        std::vector<uint32_t> tp(N + 1), tm(N + 1);
        ekf_compat::chk(e_, ekf_get_feature_layout(e_, type.data(), covpos.data()), "ekf_get_feature_layout");
        ekf_compat::chk(e_, ekf_get_map_features(e_, desc.data(), tp.data(), tm.data()), "ekf_get_map_features");
        state.removeAllFeatures();
        const double zero[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < N; ++i) {
            Descriptor32 d;
            std::memcpy(d.bytes, &desc[(size_t)i * EKF_DESC_BYTES], EKF_DESC_BYTES);
            const bool inv = type[i] == EKF_FEATURE_INVERSE_DEPTH;
            MapFeature *f = new MapFeature(zero, inv ? 6 : 3, covpos[i], d, inv ? MAPFEATURE_TYPE_INVERSE_DEPTH : MAPFEATURE_TYPE_DEPTH);
            f->timesPredicted = tp[i];
            f->timesMatched = tm[i];
            state.addFeature(f);
        }
    }
    EkfEngine *e_;
    int steps_;
};

#endif // EKF_COMPAT_H
