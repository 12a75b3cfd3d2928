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
//              matchPredictedFeatures (Matching.h:66: image in -- matcher mode B, the build's pyramid NCC matcher; an
//              overload takes the frame's keypoints + descriptors where the reference's detector/descriptor stage ran
//              on the host), ransac (1PointRansac.h:42), updateOnlyState (Update.h:42), update (Update.h:48),
//              rescueOutliers (EKF.cpp:68)
//   class      EKF { EKF(configurationFileName, outputPath), init(image), step(image), state, stateCovarianceMatrix }
//              (EKF.h:41-63) -- device-resident between steps; the file-based constructor lives in ekf_io.h.
//
// Error behaviour: the reference signals nothing (void everywhere).  Here a failing engine call throws
// ekf_compat::Error carrying the C status code; a singular S is reported instead of silently producing zeros.
//
// The free functions keep the reference's signatures and its contract (the caller's State and covariance are current
// after every call) WITHOUT moving the covariance per call (round 6, "resident mode"): the process-wide engine behind
// them (ekf_compat::Context) remembers which (State *, Matd *) it holds.  While the caller passes the same objects and
// has not written to them, nothing is uploaded; after a call that changes the filter the small state (13 + 6 N doubles)
// is copied back at once and the covariance is only MARKED stale: the first host access to it (operator[], ptr(), at(),
// a copy) pulls it from the device, a non-const access also tells the engine to upload it again next time.  EKF::step's
// own call sequence (EKF.cpp:273-531) never touches P on the host, so at N = 1000 it moves 145 MB zero times per frame
// instead of once per stage.  Jacobian vectors handed to ransac / update / updateOnlyState must be the ones
// predictCameraMeasurements returned for the same (state, P) (the reference's own contract, Update.h:47); the wrappers
// use the device-side tables of that call rather than uploading dense 2 x n matrices.
// cv::Mat: a minimal look-alike (rows, cols, step, data, channels()) so that init(const cv::Mat &), step(const cv::Mat &)
// and matchPredictedFeatures(const cv::Mat &, ...) have the reference's signatures; with -DEKF_COMPAT_HAVE_OPENCV the
// host's own <opencv2/core/core.hpp> is used instead (the wrappers only read those five members).
// The EKF class stays the efficient path: everything lives in HBM across init() / step().
#ifndef EKF_COMPAT_H
#define EKF_COMPAT_H

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ekf_engine.h"

// ------------------------------------------------------------------------------------------------------ cv::Mat
#ifdef EKF_COMPAT_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
namespace cv {
// What the boundary reads of an image: 8-bit, 1 / 3 (BGR) / 4 channels, row-major with a row stride in bytes
// (Img/FileSequenceImageGenerator.cpp:82 and android jni/EKFNative.cpp:163 deliver exactly that).  A non-owning view unless
// created with (rows, cols, channels).
class Mat {
public:
    int rows, cols;
    size_t step; // bytes per row
    unsigned char *data;
    Mat() : rows(0), cols(0), step(0), data(0), ch_(0) {}
    Mat(int r, int c, int channels) : rows(r), cols(c), step((size_t)c * channels), data(0), ch_(channels), own_((size_t)r * c * channels, 0)
    {
        data = own_.data();
    }
    Mat(int r, int c, int channels, unsigned char *pixels, size_t stepBytes = 0)
        : rows(r), cols(c), step(stepBytes ? stepBytes : (size_t)c * channels), data(pixels), ch_(channels) {}
    Mat(const Mat &o) { *this = o; }
    Mat &operator=(const Mat &o)
    {
        rows = o.rows; cols = o.cols; step = o.step; ch_ = o.ch_; own_ = o.own_;
        data = own_.empty() ? o.data : own_.data();
        return *this;
    }
    int channels() const { return ch_; }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }

private:
    int ch_;
    std::vector<unsigned char> own_;
};
} // namespace cv
#endif

// ---------------------------------------------------------------------------------------------------- matrices
class Matd;
class State;
namespace ekf_compat {
inline void pull_matrix(const Matd *m); // fills a stale matrix from the engine that holds its current value (below)
inline void matrix_destroyed(const Matd *m); // the resident engine forgets a covariance / a state whose object goes away (below)
inline void state_destroyed(const State *s);
}

// cv::Mat_<double> as the path uses it (Core/Base.h:171): an owning row-major matrix.  Resident mode (see the header): a
// matrix a wrapper left on the device is marked stale; any host access fills it first, a non-const access counts as a write.
class Matd {
public:
    int rows, cols;
    Matd() : rows(0), cols(0), gen_(0), stale_(false), src_(0), tracked_(false) {}
    Matd(int r, int c) : rows(r), cols(c), d_((size_t)r * c, 0.0), gen_(0), stale_(false), src_(0), tracked_(false) {}
    Matd(const Matd &o) : rows(0), cols(0), gen_(0), stale_(false), src_(0), tracked_(false) { *this = o; }
    ~Matd() { if (tracked_) ekf_compat::matrix_destroyed(this); }
    Matd &operator=(const Matd &o)
    {
        if (this == &o) return *this;
        o.hostRead();
        rows = o.rows; cols = o.cols; d_ = o.d_;
        ++gen_;
        stale_ = false;
        src_ = 0;
        return *this;
    }
    static Matd zeros(int r, int c) { return Matd(r, c); }
    static Matd eye(int r, int c)
    {
        Matd m(r, c);
        for (int i = 0; i < r && i < c; ++i) m[i][i] = 1.0;
        return m;
    }
    double *operator[](int i) { hostWrite(); return d_.data() + (size_t)i * cols; }
    const double *operator[](int i) const { hostRead(); return d_.data() + (size_t)i * cols; }
    double *ptr() { hostWrite(); return d_.data(); }
    const double *ptr() const { hostRead(); return d_.data(); }
    double &at(int i, int j) { hostWrite(); return d_[(size_t)i * cols + j]; }
    Matd clone() const { return *this; }

    // ---- resident mode plumbing (ekf_compat::Context, class EKF); not part of the reference's surface
    unsigned long hostGeneration() const { return gen_; }          // bumps on every possible host write
    bool staleOnHost() const { return stale_; }
    void markStale(EkfEngine *holder, int n) const                 // the device copy is newer; (n x n) once pulled
    {
        stale_ = true;
        src_ = holder;
        const_cast<Matd *>(this)->rows = const_cast<Matd *>(this)->cols = n;
    }
    EkfEngine *holder() const { return src_; }
    double *rawForPull() const                                     // storage for the pull, sized rows x cols
    {
        d_.resize((size_t)rows * cols);
        return d_.data();
    }
    void pulled() const { stale_ = false; src_ = 0; }
    void forget() const { stale_ = false; src_ = 0; }              // the holder went away with a copy nobody asked for
    void setTracked(bool t) const { tracked_ = t; }                // the resident engine holds a pointer to this object

private:
    void hostRead() const { if (stale_) ekf_compat::pull_matrix(this); }
    void hostWrite() { hostRead(); ++gen_; }
    mutable std::vector<double> d_;
    unsigned long gen_;
    mutable bool stale_;
    mutable EkfEngine *src_;
    mutable bool tracked_;
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
    ~State()
    {
        if (tracked_) ekf_compat::state_destroyed(this);
        removeAllFeatures();
    }
    void setTracked(bool t) const { tracked_ = t; } // the resident engine of the free functions holds a pointer to this object
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
        tracked_ = false;
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
    mutable bool tracked_;
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

inline void chk(EkfEngine *e, int rc, const char *what)
{
    if (rc != EKF_OK) throw Error(rc, std::string(what) + ": " + (e ? ekf_last_error(e) : ""));
}

inline void pull_matrix(const Matd *m)
{
    EkfEngine *e = m->holder();
    if (!e) { m->forget(); return; }
    const int n = ekf_state_dim(e);
    const_cast<Matd *>(m)->rows = const_cast<Matd *>(m)->cols = n;
    chk(e, ekf_get_state(e, 0, 0, m->rawForPull()), "ekf_get_state (covariance pulled on first host access)");
    m->pulled();
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

// the small part of the filter back into the caller's State; the covariance too when P != 0
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

// Replaces the ConfigurationManager singleton (Configuration/ConfigurationManager.h:45-64) for the two PODs the
// hot path reads.  One process-wide engine serves the free functions and remembers whose filter it holds.
class Context;
inline Context *&context_alive() // the singleton while it exists (objects destroyed after it must not call into it)
{
    static Context *p = 0;
    return p;
}

class Context {
public:
    static Context &instance()
    {
        static Context c;
        return c;
    }
    // the caller's objects went out of scope: the engine keeps the filter, but no pointer to them
    void forgetMatrix(const Matd *m)
    {
        if (rP_ == m) { rP_ = 0; live_ = false; }
    }
    void forgetState(const State *s)
    {
        if (rs_ == s) { rs_ = 0; live_ = false; }
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
        if (e_ && rP_ && rP_->staleOnHost() && rP_->holder() == e_) pull_matrix(rP_); // the caller's matrix stays valid
        if (e_) ekf_engine_destroy(e_);
        e_ = 0;
        live_ = false;
        track(0, 0);
        covUploads_ = 0;
        featGen_.clear();
    }
    ~Context()
    {
        release();
        context_alive() = 0;
    }

    // ---- resident mode -----------------------------------------------------------------------------------------------------
    // Makes the engine hold the caller's (state, P): nothing moves when it already does (same objects, no host write since).
    // P == 0: the stage does not read the covariance (predictMeasurementState).
    void ensure(const State &s, const Matd *P)
    {
        EkfEngine *e = engine();
        const bool same_state = live_ && rs_ == &s && snapshotEquals(s);
        const bool same_P = live_ && P != 0 && rP_ == P && pgen_ == P->hostGeneration();
        if (same_state && (P == 0 || same_P)) return;
        if (P != 0 && !same_P) {
            if (rP_ && rP_ != P && rP_->staleOnHost() && rP_->holder() == e) pull_matrix(rP_); // somebody else's matrix: hand it back first
            upload(e, s, P); // (reads P: a matrix that is stale from THIS engine is pulled by the access and uploaded again)
            ++covUploads_;
            track(&s, P);
            pgen_ = P->hostGeneration();
        } else {
            upload(e, s, 0); // the state was edited on the host (or is another object); the device covariance stays
            track(&s, rP_);
        }
        live_ = true;
        takeSnapshot(s);
        ++devGen_; // every device table is older than this filter
    }
    // after a stage that changed the filter on the device: the small state goes back now, the covariance on first access
    void changed(State &s, Matd *P)
    {
        EkfEngine *e = engine();
        download(e, s, 0);
        takeSnapshot(s);
        if (P) {
            P->markStale(e, ekf_state_dim(e));
            pgen_ = P->hostGeneration();
        }
        ++devGen_;
    }
    bool live() const { return live_; }
    long covarianceUploads() const { return covUploads_; } // how often a covariance travelled host -> device since configure()
    // device tables (prediction, Jacobian blocks, H P rows) of feature f belong to the filter as it is now?
    bool tablesCurrent(int f) const { return f >= 0 && (size_t)f < featGen_.size() && featGen_[f] == devGen_; }
    void tablesComputed(const int32_t *idx, int count, int nFeatures)
    {
        if (featGen_.size() < (size_t)nFeatures) featGen_.resize(nFeatures, 0);
        if (idx == 0)
            for (int f = 0; f < nFeatures; ++f) featGen_[f] = devGen_;
        else
            for (int k = 0; k < count; ++k) featGen_[idx[k]] = devGen_;
    }
    // Mode-B templates of the listed features, cut from `image` around uv (2 doubles each) -- this build's "descriptor" of a map
    // feature for the image-taking matchPredictedFeatures (the reference keeps a BRIEF descriptor in MapFeature::descriptor,
    // AddMapFeature.cpp:317-337)
    void captureTemplates(const cv::Mat &image, const std::vector<int32_t> &featureIndexes, const std::vector<double> &uv)
    {
        EkfEngine *e = engine();
        check(ekf_image_upload(e, image.data, image.cols, image.rows, (int)image.step, image.channels()), "ekf_image_upload");
        check(ekf_capture_templates(e, featureIndexes.data(), uv.data(), (int)featureIndexes.size()), "ekf_capture_templates");
    }

private:
    Context() : e_(0), live_(false), rs_(0), rP_(0), pgen_(0), devGen_(1), covUploads_(0) { context_alive() = this; }
    void track(const State *s, const Matd *P)
    {
        if (rs_ && rs_ != s) rs_->setTracked(false);
        if (rP_ && rP_ != P) rP_->setTracked(false);
        rs_ = s;
        rP_ = P;
        if (rs_) rs_->setTracked(true);
        if (rP_) rP_->setTracked(true);
    }
    void takeSnapshot(const State &s)
    {
        snap_.assign(s.x13(), s.x13() + 13);
        for (size_t i = 0; i < s.mapFeatures.size(); ++i) {
            const MapFeature *f = s.mapFeatures[i];
            snap_.push_back((double)f->featureType);
            snap_.insert(snap_.end(), f->position, f->position + f->positionDimension);
        }
    }
    bool snapshotEquals(const State &s) const
    {
        size_t k = 13;
        if (snap_.size() < 13 || std::memcmp(snap_.data(), s.x13(), 13 * sizeof(double)) != 0) return false;
        for (size_t i = 0; i < s.mapFeatures.size(); ++i) {
            const MapFeature *f = s.mapFeatures[i];
            if (k + 1 + f->positionDimension > snap_.size() || snap_[k] != (double)f->featureType) return false;
            if (std::memcmp(&snap_[k + 1], f->position, sizeof(double) * f->positionDimension) != 0) return false;
            k += 1 + f->positionDimension;
        }
        return k == snap_.size();
    }
    EkfEngine *e_;
    EkfEngineConfig cfg_;
    bool live_;
    const State *rs_;
    const Matd *rP_;
    unsigned long pgen_;
    std::vector<double> snap_;
    unsigned long devGen_;
    std::vector<unsigned long> featGen_;
    long covUploads_;
};

inline void matrix_destroyed(const Matd *m)
{
    if (context_alive()) context_alive()->forgetMatrix(m);
}
inline void state_destroyed(const State *s)
{
    if (context_alive()) context_alive()->forgetState(s);
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

// device-side tables for exactly the features the caller's predictions refer to -- recomputed only for those whose tables
// are older than the filter the engine holds (after predictCameraMeasurements on the same filter: none)
inline void refreshTables(EkfEngine *e, const VectorImageFeaturePrediction &preds)
{
    Context &c = Context::instance();
    std::vector<int32_t> idx;
    for (size_t i = 0; i < preds.size(); ++i)
        if (!c.tablesCurrent(preds[i]->featureIndex)) idx.push_back(preds[i]->featureIndex);
    int n = 0;
    if (!idx.empty()) {
        c.check(ekf_predict_measurements(e, idx.data(), (int)idx.size(), 0, &n, 0, 0), "ekf_predict_measurements");
        c.tablesComputed(idx.data(), (int)idx.size(), ekf_num_features(e));
    }
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
                double *row = (*J)[r];
                for (int c = 0; c < 13; ++c) row[c] = Hs[(size_t)26 * k + r * 13 + c];
                for (int c = 0; c < f->positionDimension; ++c) row[f->covarianceMatrixPos + c] = Hf[(size_t)12 * k + r * 6 + c];
            }
            jac->push_back(J);
        }
    }
}

inline void emitMatches(const std::vector<EkfMatch> &m, int M, const uint8_t *descriptors, VectorFeatureMatch &matches)
{
    for (int i = 0; i < M; ++i) {
        FeatureMatch *fm = new FeatureMatch;
        fm->featureIndex = m[i].featureIndex;
        fm->imagePos[0] = m[i].imagePos[0];
        fm->imagePos[1] = m[i].imagePos[1];
        fm->distance = m[i].distance;
        if (descriptors && m[i].keypointIndex >= 0)
            std::memcpy(fm->imagePosDescriptor.bytes, descriptors + (size_t)EKF_DESC_BYTES * m[i].keypointIndex, EKF_DESC_BYTES);
        matches.push_back(fm);
    }
}

} // namespace ekf_compat

// ------------------------------------------------------------------------------------- the reference's functions
inline void stateAndCovariancePrediction(State &state, Matd &covarianceMatrix)
{
    ekf_compat::Context &c = ekf_compat::Context::instance();
    c.ensure(state, &covarianceMatrix);
    c.check(ekf_predict(c.engine()), "ekf_predict");
    c.changed(state, &covarianceMatrix);
}

inline void predictCameraMeasurements(const State &state, const Matd &predictedStateCovariance, const VectorMapFeature &features,
                                      const std::vector<int> &featureIndexes, VectorImageFeaturePrediction &predictedDistortedFeatures,
                                      VectorMatd &predictedFeatureJacobians, VectorMapFeature &notPredictedFeatures)
{
    ekf_compat::Context &c = ekf_compat::Context::instance();
    c.ensure(state, &predictedStateCovariance);
    EkfEngine *e = c.engine();
    const int cnt = (int)features.size();
    if (cnt == 0) return;
    std::vector<EkfPrediction> p(cnt);
    std::vector<double> Hs((size_t)26 * cnt), Hf((size_t)12 * cnt);
    std::vector<int32_t> idx(featureIndexes.begin(), featureIndexes.end());
    int n = 0;
    c.check(ekf_predict_measurements(e, idx.empty() ? 0 : idx.data(), (int)idx.size(), p.data(), &n, Hs.data(), Hf.data()),
            "ekf_predict_measurements");
    c.tablesComputed(idx.empty() ? 0 : idx.data(), (int)idx.size(), (int)state.mapFeatures.size());
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
    ekf_compat::Context &c = ekf_compat::Context::instance();
    c.ensure(state, 0);
    EkfEngine *e = c.engine();
    const int N = (int)state.mapFeatures.size();
    if (features.empty()) return;
    std::vector<EkfPrediction> p(N + 1);
    int n = 0;
    c.check(ekf_predict_measurement_state(e, p.data(), &n), "ekf_predict_measurement_state");
    std::vector<int> slot(N, -1);
    for (int k = 0; k < n; ++k) slot[p[k].featureIndex] = k;
    for (size_t i = 0; i < features.size(); ++i) {
        const int fi = featureIndexes.empty() ? (int)i : featureIndexes[i];
        if (slot[fi] >= 0) predictedFeatures.push_back(new ImageFeaturePrediction(fi, p[slot[fi]].imagePos));
        else notPredictedFeatures.push_back(features[i]);
    }
}

// matchPredictedFeatures(image, features, vectorfeaturePrediction, matches) -- the reference's signature (Matching.h:66).  The
// predictions are those of the last predictCameraMeasurements call on the filter the engine holds (their gates live in the
// device tables keyed by featureIndex); the image is matched by matcher mode B: zero-mean NCC of each feature's 11 x 11 template
// inside its predicted ellipse over a 3-level pyramid (templates: Context::captureTemplates).  Matches carry integer pixel
// positions and no descriptor.
inline void matchPredictedFeatures(const cv::Mat &image, const VectorMapFeature &features,
                                   const VectorImageFeaturePrediction &vectorfeaturePrediction, VectorFeatureMatch &matches)
{
    ekf_compat::Context &c = ekf_compat::Context::instance();
    if (!c.live()) throw ekf_compat::Error(EKF_ERR_INVALID_ARG, "matchPredictedFeatures: no filter on the device (call predictCameraMeasurements first)");
    if (vectorfeaturePrediction.empty()) return;
    EkfEngine *e = c.engine();
    c.check(ekf_image_upload(e, image.data, image.cols, image.rows, (int)image.step, image.channels()), "ekf_image_upload");
    std::vector<EkfMatch> m(features.size() + 1);
    int M = 0;
    c.check(ekf_match_ncc(e, m.data(), &M), "ekf_match_ncc");
    ekf_compat::emitMatches(m, M, 0, matches);
}

// ... and with the frame's keypoints + descriptors where the reference's detector / descriptor extractor ran
// (Matching.cpp:188-210 is host-side OpenCV code outside the path): the ellipse gate, the descriptor distance and the 2-best rule
// of Matching.cpp:217-262 against the same device tables
inline void matchPredictedFeatures(const FrameKeypoints &frame, const VectorMapFeature &features,
                                   const VectorImageFeaturePrediction &vectorfeaturePrediction, VectorFeatureMatch &matches)
{
    ekf_compat::Context &c = ekf_compat::Context::instance();
    if (!c.live()) throw ekf_compat::Error(EKF_ERR_INVALID_ARG, "matchPredictedFeatures: no filter on the device (call predictCameraMeasurements first)");
    if (vectorfeaturePrediction.empty()) return;
    std::vector<EkfMatch> m(features.size() + 1);
    int M = 0;
    c.check(ekf_match(c.engine(), frame.keypoints.data(), frame.descriptors.data(), (int)frame.keypoints.size(), m.data(), &M), "ekf_match");
    ekf_compat::emitMatches(m, M, frame.descriptors.data(), matches);
}

// (rounds 1-5 form: the filter passed explicitly; predicts every feature again before matching)
inline void matchPredictedFeatures(const FrameKeypoints &frame, const State &state, const Matd &covariance,
                                   const VectorImageFeaturePrediction &vectorfeaturePrediction, VectorFeatureMatch &matches)
{
    ekf_compat::Context &c = ekf_compat::Context::instance();
    c.ensure(state, &covariance);
    int n = 0;
    c.check(ekf_predict_measurements(c.engine(), 0, 0, 0, &n, 0, 0), "ekf_predict_measurements");
    c.tablesComputed(0, 0, (int)state.mapFeatures.size());
    matchPredictedFeatures(frame, state.mapFeatures, vectorfeaturePrediction, matches);
}

inline void ransac(const State &state, Matd &covariance, const VectorImageFeaturePrediction &predictedMatchedFeatures,
                   const VectorMatd &predictedMatchedJacobians, const VectorFeatureMatch &matches, VectorFeatureMatch &inlierMatches,
                   VectorImageFeaturePrediction &inlierPredictions, VectorMatd &inlierJacobians, VectorFeatureMatch &outlierMatches)
{
    if (matches.empty()) return;
    ekf_compat::Context &c = ekf_compat::Context::instance();
    c.ensure(state, &covariance);
    EkfEngine *e = c.engine();
    ekf_compat::refreshTables(e, predictedMatchedFeatures);
    std::vector<EkfMatch> m = ekf_compat::packMatches(matches);
    std::vector<uint8_t> mask(m.size());
    c.check(ekf_ransac(e, m.data(), (int)m.size(), mask.data(), 0), "ekf_ransac");
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
    ekf_compat::Context &c = ekf_compat::Context::instance();
    c.ensure(state, &covariance);
    EkfEngine *e = c.engine();
    ekf_compat::refreshTables(e, predictedDistortedFeatures);
    std::vector<EkfMatch> m = ekf_compat::packMatches(measurementMatchedFeatures);
    c.check(ekf_update(e, m.data(), (int)m.size()), "ekf_update");
    c.changed(state, &covariance);
}

inline void updateOnlyState(const VectorImageFeaturePrediction &predictedDistortedFeatures, const VectorFeatureMatch &measurementMatchedFeatures,
                            const VectorMatd &predictedFeatureJacobians, State &state, Matd &covariance)
{
    (void)predictedFeatureJacobians;
    if (measurementMatchedFeatures.empty()) return;
    ekf_compat::Context &c = ekf_compat::Context::instance();
    c.ensure(state, &covariance);
    EkfEngine *e = c.engine();
    ekf_compat::refreshTables(e, predictedDistortedFeatures);
    std::vector<EkfMatch> m = ekf_compat::packMatches(measurementMatchedFeatures);
    c.check(ekf_update_only_state(e, m.data(), (int)m.size()), "ekf_update_only_state");
    c.changed(state, 0);
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
// EKF.h:41-63.  Two ways in:
//   * the reference's own:  EKF(configurationFileName, outputPath), init(image), step(image)  -- the YAML configuration, output.yml /
//     log.txt, new-feature detection and map management of EKF::step around the device-resident engine, matcher mode B.  These
//     three members are DEFINED in ekf_io.h (configuration loader, image / YAML I/O): include it where they are used.
//   * the PODs:  EKF(cam, par, maxFeatures), init(seed state, P), step(frame keypoints)  -- for a host that keeps its own
//     configuration and detector.
// state and stateCovarianceMatrix are the reference's public attributes: after step(image) `state` is current and
// `stateCovarianceMatrix` is current on first access (it is pulled from the device when somebody reads it, not every frame);
// after step(keypoints) call syncToHost() for `state`.
namespace ekf_compat { class ImageEKF; }

class EKF {
public:
    EKF(const char *configurationFileName, const char *outputPath); // EKF.h:44   (ekf_io.h)
    void init(const cv::Mat &image);                                // EKF.h:47   (ekf_io.h)
    void step(const cv::Mat &image);                                // EKF.h:48   (ekf_io.h)

    EKF(const EkfCamera &cam, const EkfParams &par, int maxFeatures, int precision = EKF_PRECISION_F64)
        : e_(0), steps_(0), drv_(0), drvDelete_(0)
    {
        std::memset(&last_, 0, sizeof(last_));
        EkfEngineConfig cfg = EkfEngineConfig();
        cfg.cam = cam; cfg.par = par; cfg.max_features = maxFeatures; cfg.precision = precision; cfg.device = -1;
        const int rc = ekf_engine_create(&cfg, &e_);
        if (rc != EKF_OK) throw ekf_compat::Error(rc, "ekf_engine_create failed (no MI355X?)");
    }
    ~EKF()
    {
        stateCovarianceMatrix.forget(); // (nobody can read it any more: no pull from an engine that is going away)
        if (drv_) drvDelete_(drv_);     // the image driver owns its engine
        else if (e_) ekf_engine_destroy(e_);
    }
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
        stateCovarianceMatrix.markStale(e_, ekf_state_dim(e_));
        last_ = info;
        return info;
    }
    const EkfStepInfo &lastStepInfo() const { return last_; }
    void syncToHost()
    {
        refreshLayout();
        (void)static_cast<const Matd &>(stateCovarianceMatrix).ptr(); // pulled here if the device copy is newer
        ekf_compat::download(e_, state, 0);
    }
    // ---- map management on the device (EKF.cpp:574-612).  Detection of new features stays on the
    // host: the caller passes the distorted pixel + descriptor of each new ImageFeatureMeasurement.
    void addFeaturesToStateAndCovariance(const std::vector<double> &uv, const std::vector<uint8_t> &descriptors)
    { // AddMapFeature.cpp:354
        ekf_compat::chk(e_, ekf_add_features(e_, uv.data(), descriptors.empty() ? 0 : descriptors.data(), (int)(uv.size() / 2)),
                        "ekf_add_features");
        stateCovarianceMatrix.markStale(e_, ekf_state_dim(e_));
    }
    int removeBadMapFeatures()
    { // MapManagement.cpp:279
        int k = 0;
        ekf_compat::chk(e_, ekf_remove_bad_features(e_, &k), "ekf_remove_bad_features");
        stateCovarianceMatrix.markStale(e_, ekf_state_dim(e_));
        return k;
    }
    void removeFeaturesFromStateAndCovariance(const std::vector<int32_t> &ascendingFeatureIndexes)
    { // MapManagement.cpp:212
        ekf_compat::chk(e_, ekf_remove_features(e_, ascendingFeatureIndexes.data(), (int)ascendingFeatureIndexes.size()),
                        "ekf_remove_features");
        stateCovarianceMatrix.markStale(e_, ekf_state_dim(e_));
    }
    int convertMapFeaturesInverseDepthToDepth()
    { // MapManagement.cpp:494
        int k = -1;
        ekf_compat::chk(e_, ekf_convert_inverse_depth_to_depth(e_, &k), "ekf_convert_inverse_depth_to_depth");
        stateCovarianceMatrix.markStale(e_, ekf_state_dim(e_));
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
    EkfStepInfo last_;
    ekf_compat::ImageEKF *drv_;                   // the reference-style constructor's driver (ekf_io.h), or null
    void (*drvDelete_)(ekf_compat::ImageEKF *);
};

#endif // EKF_COMPAT_H
