// ekf_io.h -- the pieces either side of the hot path that make the reference's sample command line work unchanged
//     EKF config.yml imgdir/ outdir/        (kalmanFilter/samples/EKF/main.cpp:45-160)
// on top of the MI355X engine's C ABI (matcher mode B: the frame goes in, no OpenCV detector):
//   * loadConfiguration        Configuration/ConfigurationManager.cpp:74-111 + the four *Configuration readers
//                              (OpenCV FileStorage YAML 1.0, values stored as quoted strings)
//   * FileSequenceImageGenerator  ImageGenerator/FileSequenceImageGenerator.cpp:61-97  ("%s%s%05d.%s", BGR like cv::imread)
//   * OutputWriter             the output.yml the reference writes through cv::FileStorage
//                              (1PointRansacEKF/EKF.cpp:133,262-266,291,344,410-416,437,513-516,539,618-628;
//                               State.cpp:339-360): same keys, same nesting, readable by cv::FileStorage
//   * ImageEKF                 EKF(config, out) / init(image) / step(image)   1PointRansacEKF/EKF.h:41-63, with the
//                              map-management block of EKF.cpp:572-612 on the device
// C++11, header-only, needs zlib (-lz) for PNG.  SURVEY.md 8(f)-3 and 8(f)-4.
#ifndef EKF_IO_H
#define EKF_IO_H

#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ekf_engine.h"

namespace ekf_compat {

// ------------------------------------------------------------------------------------------- configuration
// Fields of ExtendedKalmanFilterParameters (ExtendedKalmanFilterParameters.h:37-76) the DRIVER reads; the ones the
// engine reads are in EkfParams.
struct RunParameters {
    int reserveFeaturesDepth, reserveFeaturesInvDepth;
    int maxMapSize, maxMapFeaturesCount;
    bool alwaysRemoveUnseenMapFeatures;
    int mapManagementFrequency;
    int detectNewFeaturesImageAreasDivideTimes;
    double detectNewFeaturesImageMaskEllipseSize;
    int minMatchesPerImage;
    RunParameters()
        : reserveFeaturesDepth(1024), reserveFeaturesInvDepth(1024), maxMapSize(0), maxMapFeaturesCount(0),
          alwaysRemoveUnseenMapFeatures(false), mapManagementFrequency(1), detectNewFeaturesImageAreasDivideTimes(2),
          detectNewFeaturesImageMaskEllipseSize(10.0), minMatchesPerImage(60) {}
};

struct ConfigNode {
    std::string value;
    std::map<std::string, ConfigNode> children;
    const ConfigNode *find(const std::string &k) const
    {
        std::map<std::string, ConfigNode>::const_iterator it = children.find(k);
        return it == children.end() ? 0 : &it->second;
    }
};

inline std::string trimQuotes(std::string s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    s = s.substr(a, b - a + 1);
    if (s.size() >= 2 && ((s[0] == '"' && s[s.size() - 1] == '"') || (s[0] == '\'' && s[s.size() - 1] == '\''))) s = s.substr(1, s.size() - 2);
    return s;
}

// The subset of YAML 1.0 cv::FileStorage emits for nested maps of scalars: "key: value" lines, nesting by indentation.
inline bool parseYamlMaps(std::istream &in, ConfigNode &root, std::string *err)
{
    std::vector<std::pair<int, ConfigNode *> > stack;
    stack.push_back(std::make_pair(-1, &root));
    std::string line;
    int ln = 0;
    while (std::getline(in, line)) {
        ++ln;
        const size_t hash = line.find('#');
        if (hash != std::string::npos && line.find('"') > hash) line = line.substr(0, hash);
        const size_t first = line.find_first_not_of(" \t\r");
        if (first == std::string::npos || line[first] == '%' || line.compare(first, 3, "---") == 0) continue;
        const size_t colon = line.find(':', first);
        if (colon == std::string::npos) {
            if (err) { std::ostringstream o; o << "line " << ln << ": expected 'key: value'"; *err = o.str(); }
            return false;
        }
        const int indent = (int)first;
        while (stack.size() > 1 && stack.back().first >= indent) stack.pop_back();
        ConfigNode &node = stack.back().second->children[trimQuotes(line.substr(first, colon - first))];
        node.value = trimQuotes(line.substr(colon + 1));
        stack.push_back(std::make_pair(indent, &node));
    }
    return true;
}

inline double cfgNumber(const ConfigNode &n, const char *key, double def, bool *missing = 0)
{
    const ConfigNode *c = n.find(key);
    if (!c || c->value.empty()) { if (missing) *missing = true; return def; }
    return std::strtod(c->value.c_str(), 0);
}

// ConfigurationManager::loadConfigFromFile (Configuration/ConfigurationManager.cpp:74-111): RunConfiguration names the
// profile to take from each of the sections.
inline bool loadConfiguration(const std::string &path, EkfCamera &cam, EkfParams &par, RunParameters &run, std::string *err = 0)
{
    std::ifstream in(path.c_str());
    if (!in) { if (err) *err = "cannot open " + path; return false; }
    ConfigNode root;
    if (!parseYamlMaps(in, root, err)) return false;
    const ConfigNode *rc = root.find("RunConfiguration");
    if (!rc) { if (err) *err = "RunConfiguration section missing"; return false; }
    const ConfigNode *ekfName = rc->find("ExtendedKalmanFilter"), *camName = rc->find("CameraCalibration");
    const ConfigNode *ekfSec = root.find("ExtendedKalmanFilter"), *camSec = root.find("CameraCalibration");
    if (!ekfName || !camName || !ekfSec || !camSec) { if (err) *err = "ExtendedKalmanFilter / CameraCalibration sections missing"; return false; }
    const ConfigNode *k = ekfSec->find(ekfName->value), *c = camSec->find(camName->value);
    if (!k || !c) { if (err) *err = "profile '" + ekfName->value + "' or '" + camName->value + "' not found"; return false; }
    bool miss = false;
    cam.pixelsX = (int)cfgNumber(*c, "PixelsX", 0, &miss); cam.pixelsY = (int)cfgNumber(*c, "PixelsY", 0, &miss);
    cam.fx = cfgNumber(*c, "FX", 0, &miss); cam.fy = cfgNumber(*c, "FY", 0, &miss);
    cam.k1 = cfgNumber(*c, "K1", 0, &miss); cam.k2 = cfgNumber(*c, "K2", 0, &miss);
    cam.cx = cfgNumber(*c, "CX", 0, &miss); cam.cy = cfgNumber(*c, "CY", 0, &miss);
    cam.dx = cfgNumber(*c, "DX", 0, &miss); cam.dy = cfgNumber(*c, "DY", 0, &miss);
    cam.pixelErrorX = cfgNumber(*c, "PixelErrorX", 1, &miss); cam.pixelErrorY = cfgNumber(*c, "PixelErrorY", 1, &miss);
    cam.angularVisionX = cfgNumber(*c, "AngularVisionX", 0, &miss); cam.angularVisionY = cfgNumber(*c, "AngularVisionY", 0, &miss);
    par.initInvDepthRho = cfgNumber(*k, "InitInvDepthRho", 1, &miss);
    par.initLinearAccelSD = cfgNumber(*k, "InitLinearAccelSD", 0, &miss);
    par.initAngularAccelSD = cfgNumber(*k, "InitAngularAccelSD", 0, &miss);
    par.linearAccelSD = cfgNumber(*k, "LinearAccelSD", 0, &miss);
    par.angularAccelSD = cfgNumber(*k, "AngularAccelSD", 0, &miss);
    par.inverseDepthRhoSD = cfgNumber(*k, "InverseDepthRhoSD", 1, &miss);
    par.matchingCompCoefSecondBestVSFirst = cfgNumber(*k, "MatchingCompCoefSecondBestVSFirst", 1, &miss);
    par.ransacThresholdPredictDistance = cfgNumber(*k, "RansacThresholdPredictDistance", 1, &miss);
    par.ransacAllInliersProbability = cfgNumber(*k, "RansacAllInliersProbability", 0.99, &miss);
    par.ransacChi2Threshold = cfgNumber(*k, "RansacChi2Threshold", EKF_CHISQ_95_2, &miss);
    par.goodFeatureMatchingPercent = cfgNumber(*k, "GoodFeatureMatchingPercent", 0.5, &miss);
    par.inverseDepthLinearityIndexThreshold = cfgNumber(*k, "InverseDepthLinearityIndexThreshold", 0.1, &miss);
    if (miss) { if (err) *err = "a camera or filter parameter is missing in " + path; return false; }
    run.reserveFeaturesDepth = (int)cfgNumber(*k, "ReserveFeaturesDepth", 1024);
    run.reserveFeaturesInvDepth = (int)cfgNumber(*k, "ReserveFeaturesInvDepth", 1024);
    run.maxMapSize = (int)cfgNumber(*k, "MaxMapSize", 0);
    run.maxMapFeaturesCount = (int)cfgNumber(*k, "MaxMapFeaturesCount", 0);
    const ConfigNode *aru = k->find("AlwaysRemoveUnseenMapFeatures");
    run.alwaysRemoveUnseenMapFeatures = aru && (aru->value == "true" || aru->value == "1");
    run.mapManagementFrequency = (int)cfgNumber(*k, "MapManagementFrequency", 1);
    run.detectNewFeaturesImageAreasDivideTimes = (int)cfgNumber(*k, "DetectNewFeaturesImageAreasDivideTimes", 2);
    run.detectNewFeaturesImageMaskEllipseSize = cfgNumber(*k, "DetectNewFeaturesImageMaskEllipseSize", 10);
    run.minMatchesPerImage = (int)cfgNumber(*k, "MinMatchesPerImage", 60);
    return true;
}

// --------------------------------------------------------------------------------------------------- images
struct Image {
    int width, height, channels; // channels 1 (gray) or 3 (B G R, cv::imread order)
    std::vector<uint8_t> data;
    Image() : width(0), height(0), channels(0) {}
    bool empty() const { return data.empty(); }
};

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// PNG (8- or 16-bit, gray / gray+alpha / RGB / RGBA / palette, non-interlaced) -> gray or BGR bytes
inline bool readPng(const std::string &path, Image &img, std::string *err = 0)
{
    img = Image();
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8) != 0) { if (err) *err = path + ": not a PNG"; return false; }
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const std::string type((const char *)&buf[pos + 4], 4);
        const uint8_t *d = &buf[pos + 8];
        if (pos + 12 + len > buf.size()) break;
        if (type == "IHDR" && len >= 13) { w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (type == "PLTE") palette.assign(d, d + len);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "IEND") break;
        pos += 12 + len;
    }
    if (!w || !h || w > 16384 || h > 16384 || interlace || (depth != 8 && depth != 16)) { if (err) *err = path + ": unsupported PNG variant"; return false; }
    const int spp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!spp) { if (err) *err = path + ": bad colour type"; return false; }
    const size_t bpp = (size_t)spp * depth / 8, stride = bpp * w;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf outLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outLen, idat.data(), (uLong)idat.size()) != Z_OK || outLen != raw.size()) {
        if (err) *err = path + ": inflate failed";
        return false;
    }
    std::vector<uint8_t> pix(stride * h);
    for (uint32_t y = 0; y < h; ++y) { // undo the per-row filters
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *dst = &pix[stride * y];
        const uint8_t *up = y ? &pix[stride * (y - 1)] : 0;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = src[i];
            if (ft == 1) v += a;
            else if (ft == 2) v += b;
            else if (ft == 3) v += (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            dst[i] = (uint8_t)v;
        }
    }
    const bool gray = ctype == 0 || ctype == 4;
    img.width = (int)w; img.height = (int)h; img.channels = gray ? 1 : 3;
    img.data.resize((size_t)w * h * img.channels);
    const size_t step = depth / 8;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t *p = &pix[i * bpp];
        if (gray) img.data[i] = p[0];
        else if (ctype == 3) {
            const size_t q = (size_t)p[0] * 3;
            if (q + 2 >= palette.size()) { if (err) *err = path + ": palette index out of range"; return false; }
            img.data[3 * i] = palette[q + 2]; img.data[3 * i + 1] = palette[q + 1]; img.data[3 * i + 2] = palette[q];
        } else {
            img.data[3 * i] = p[2 * step]; img.data[3 * i + 1] = p[step]; img.data[3 * i + 2] = p[0];
        }
    }
    return true;
}

// gray or BGR image -> 8-bit PNG (filter 0): cv::imwrite of the per-frame prediction image (EKF.cpp:300-302), tools, tests
inline bool writePng(const std::string &path, const Image &img)
{
    const int ch = img.channels;
    if ((ch != 1 && ch != 3) || img.data.size() != (size_t)img.width * img.height * ch) return false;
    const size_t stride = (size_t)img.width * ch;
    std::vector<uint8_t> raw((stride + 1) * img.height);
    for (int y = 0; y < img.height; ++y) {
        raw[(stride + 1) * y] = 0;
        for (int x = 0; x < img.width; ++x)
            for (int c = 0; c < ch; ++c) raw[(stride + 1) * y + 1 + (size_t)x * ch + c] = img.data[((size_t)y * img.width + x) * ch + (ch == 3 ? 2 - c : c)];
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f) return false;
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    f.write((const char *)sig, 8);
    struct W {
        static void chunk(std::ofstream &o, const char *type, const uint8_t *d, uint32_t n)
        {
            uint8_t hdr[8] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
            o.write((const char *)hdr, 8);
            if (n) o.write((const char *)d, n);
            uLong crc = crc32(0L, hdr + 4, 4);
            if (n) crc = crc32(crc, d, n);
            uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
            o.write((const char *)c, 4);
        }
    };
    uint8_t ihdr[13] = {(uint8_t)(img.width >> 24), (uint8_t)(img.width >> 16), (uint8_t)(img.width >> 8), (uint8_t)img.width,
                        (uint8_t)(img.height >> 24), (uint8_t)(img.height >> 16), (uint8_t)(img.height >> 8), (uint8_t)img.height,
                        8, (uint8_t)(ch == 1 ? 0 : 2), 0, 0, 0};
    W::chunk(f, "IHDR", ihdr, 13);
    W::chunk(f, "IDAT", z.data(), (uint32_t)zlen);
    W::chunk(f, "IEND", 0, 0);
    return (bool)f;
}

// FileSequenceImageGenerator (ImageGenerator/FileSequenceImageGenerator.cpp:61-97): "<path><prefix>%05d.<ext>"
class FileSequenceImageGenerator {
public:
    FileSequenceImageGenerator(const std::string &path, const std::string &prefix, const std::string &ext, int begin, int end)
        : path_(path), prefix_(prefix), ext_(ext), index_(begin), end_(end) {}
    void init() {}
    // an empty image ends the sequence (past the end index, or the file is missing)
    Image getNextImage()
    {
        Image img;
        if (index_ > end_) return img;
        char name[32];
        std::snprintf(name, sizeof(name), "%05d", index_);
        const std::string file = path_ + prefix_ + name + "." + ext_;
        std::string err;
        if (!readPng(file, img, &err)) std::fprintf(stderr, "Unable to read image: %s\n", file.c_str());
        ++index_;
        return img;
    }
    int nextIndex() const { return index_; }

private:
    std::string path_, prefix_, ext_;
    int index_, end_;
};

// ------------------------------------------------------------------------------------------- prediction image
// drawPrediction (Gui/Draw.cpp:266-310): the frame with, per predicted feature, a 5-pixel cross (drawPoint :66-70), the
// outline of its uncertainty ellipse (drawUncertaintyEllipse2D :42-64: centre and semi-axes truncated to int) and its
// map index; inverse-depth features red, depth features green (BGR).  The reference renders the index with
// cv::putText (Hershey plain, scale 1.3); this build draws it with its own 3x5 digit glyphs -- the one stated deviation.
struct Bgr { uint8_t b, g, r; };

inline void putPixel(Image &im, int x, int y, Bgr c)
{
    if (x < 0 || y < 0 || x >= im.width || y >= im.height) return;
    uint8_t *p = &im.data[((size_t)y * im.width + x) * 3];
    p[0] = c.b; p[1] = c.g; p[2] = c.r;
}

inline void drawLine(Image &im, int x0, int y0, int x1, int y1, Bgr c)
{   // Bresenham, 8-connected (cv::line with the default line type)
    const int dx = std::abs(x1 - x0), sx = x0 < x1 ? 1 : -1, dy = -std::abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
    int e = dx + dy;
    for (;;) {
        putPixel(im, x0, y0, c);
        if (x0 == x1 && y0 == y1) break;
        const int e2 = 2 * e;
        if (e2 >= dy) { e += dy; x0 += sx; }
        if (e2 <= dx) { e += dx; y0 += sy; }
    }
}

// matrix2x2ToUncertaintyEllipse2D (Core/EKFMath.cpp:271-298): cv::eigen of the symmetric 2x2 (cyclic Jacobi,
// eigenvalues descending, eigenvectors as rows), semi-axes 2 sqrt(lambda chi2_95) as float, angle atan(V10 / V00)
inline void covarianceToEllipse(const double S[4], float axes[2], double *angle)
{
    double A01 = S[1], W0 = S[0], W1 = S[3];
    double V[4] = {1, 0, 0, 1};
    for (int it = 0; it < 120; ++it) {
        const double p = A01;
        if (std::fabs(p) <= 2.220446049250313e-16) break;
        const double y = (W1 - W0) * 0.5;
        double t = std::fabs(y) + std::hypot(p, y);
        double s = std::hypot(p, t);
        const double c = t / s;
        s = p / s;
        t = (p / t) * p;
        if (y < 0) { s = -s; t = -t; }
        A01 = 0;
        W0 -= t;
        W1 += t;
        for (int i = 0; i < 2; ++i) {
            const double a0 = V[i], b0 = V[2 + i];
            V[i] = a0 * c - b0 * s;
            V[2 + i] = a0 * s + b0 * c;
        }
    }
    if (W0 < W1) {
        std::swap(W0, W1);
        for (int i = 0; i < 2; ++i) std::swap(V[i], V[2 + i]);
    }
    axes[0] = (float)(2.0 * std::sqrt(W0 * EKF_CHISQ_95_2));
    axes[1] = (float)(2.0 * std::sqrt(W1 * EKF_CHISQ_95_2));
    *angle = std::atan(V[2] / V[0]);
}

inline void drawEllipseOutline(Image &im, int cx, int cy, int aw, int ah, double angleRad, Bgr c)
{   // cv::ellipse outline: the polygon of ellipse2Poly (5-degree steps), closed
    const double ca = std::cos(angleRad), sa = std::sin(angleRad);
    int px = 0, py = 0;
    for (int k = 0; k <= 72; ++k) {
        const double t = k * (3.14159265358979323846 / 36.0);
        const double ex = aw * std::cos(t), ey = ah * std::sin(t);
        const int x = (int)std::lrint(cx + ex * ca - ey * sa), y = (int)std::lrint(cy + ex * sa + ey * ca);
        if (k > 0) drawLine(im, px, py, x, y, c);
        px = x; py = y;
    }
}

inline void drawNumber(Image &im, int x, int y, int value, Bgr c)
{   // 3x5 glyphs, 2x magnified, baseline at y (the text origin of putText is the bottom-left corner)
    static const uint16_t glyph[10] = {075557, 022222, 071747, 071717, 055711, 074717, 074757, 071111, 075757, 075717};
    char buf[16];
    std::snprintf(buf, sizeof(buf), "%d", value);
    for (int i = 0; buf[i]; ++i) {
        const uint16_t g = glyph[buf[i] - '0'];
        for (int r = 0; r < 5; ++r)
            for (int q = 0; q < 3; ++q)
                if (g >> (14 - 3 * r - q) & 1)
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) putPixel(im, x + 8 * i + 2 * q + a, y - 10 + 2 * r + b, c);
    }
}

inline void drawPrediction(const Image &image, const EkfPrediction *preds, int nPreds, const int32_t *featureType, Image &result)
{
    result.width = image.width; result.height = image.height; result.channels = 3;
    result.data.resize((size_t)image.width * image.height * 3);
    for (size_t i = 0; i < (size_t)image.width * image.height; ++i)
        for (int ch = 0; ch < 3; ++ch) result.data[3 * i + ch] = image.channels == 3 ? image.data[3 * i + ch] : image.data[i];
    const Bgr green = {0, 255, 0}, red = {0, 0, 255}, text = {0, 255, 255}, shadow = {150, 150, 150};
    for (int i = 0; i < nPreds; ++i) {
        const EkfPrediction &p = preds[i];
        const bool inv = featureType[p.featureIndex] == EKF_FEATURE_INVERSE_DEPTH;
        const Bgr col = inv ? red : green;
        const int x = (int)p.imagePos[0], y = (int)p.imagePos[1];
        drawLine(result, x, y - 2, x, y + 2, col);
        drawLine(result, x + 2, y, x - 2, y, col);
        float axes[2];
        double angle;
        covarianceToEllipse(p.covarianceMatrix, axes, &angle);
        // centre: cv::Point(Point2f) truncates the FLOAT copy of the position; axes: cv::Size(int, int) from floats
        drawEllipseOutline(result, (int)(float)p.imagePos[0], (int)(float)p.imagePos[1], (int)std::min(axes[0], 9999999.f),
                           (int)std::min(axes[1], 9999999.f), angle, col);
        drawNumber(result, (int)std::lrint(p.imagePos[0] + 1), (int)std::lrint(p.imagePos[1] + 1), p.featureIndex, shadow);
        drawNumber(result, (int)std::lrint(p.imagePos[0]), (int)std::lrint(p.imagePos[1]), p.featureIndex, text);
    }
}

// ------------------------------------------------------------------------------------------------- log.txt
// State::showDetailed (State.cpp:229-258) = operator<<(State) (:371-400) + one line per map feature
// (operator<<(MapFeature), MapFeature.cpp:130-145), default ostream formatting (6 significant digits) as in the
// reference.
inline void showDetailed(std::ostream &os, const double x[13], int N, const double *featurePos, const int32_t *featureType,
                         const uint32_t *timesPredicted, const uint32_t *timesMatched)
{
    const double q0 = x[3], q1 = x[4], q2 = x[5], q3 = x[6];
    const double ang[3] = {std::atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2)), // quaterionToAngles, EKFMath.cpp:355-365
                           std::asin(2 * (q0 * q2 - q3 * q1)),
                           std::atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3))};
    os << "Posicion de la camara: " << x[0] << ", " << x[1] << ", " << x[2] << std::endl;
    os << "Orientacion(cuaternions): " << x[3] << ", " << x[4] << ", " << x[5] << ", " << x[6] << std::endl;
    os << "Orientacion en angulos eulerianos: " << ang[0] << ", " << ang[1] << ", " << ang[2] << std::endl;
    os << "Velocidad lineal (con respecto al mundo): " << x[7] << ", " << x[8] << ", " << x[9] << std::endl;
    os << "Velocidad angular (con respecto a la camara): " << x[10] << ", " << x[11] << ", " << x[12] << std::endl;
    os << "Cantidad de features en el mapa: " << N << std::endl;
    os << std::endl;
    os << "Map Features (" << N << "):" << std::endl;
    for (int i = 0; i < N; ++i) {
        const int d = featureType[i] == EKF_FEATURE_INVERSE_DEPTH ? 6 : 3;
        os << i << ": " << featurePos[6 * (size_t)i];
        for (int k = 1; k < d; ++k) os << ", " << featurePos[6 * (size_t)i + k];
        os << " (" << timesMatched[i] << "/" << timesPredicted[i] << ")" << std::endl;
    }
}

// --------------------------------------------------------------------------------------------- output.yml
// cv::FileStorage YAML 1.0 as OpenCV 2.4 writes it: "%YAML:1.0" header, 3-space indentation, reals as "%.16e" (or
// "<int>." when integral), matrices as !!opencv-matrix maps with a flow-sequence data field.
class OutputWriter {
public:
    OutputWriter() : depth_(0) {}
    bool open(const std::string &file)
    {
        out_.open(file.c_str());
        if (!out_) return false;
        out_ << "%YAML:1.0\n";
        return true;
    }
    bool isOpened() const { return out_.is_open(); }
    void release() { if (out_.is_open()) out_.close(); }
    void beginMap(const std::string &name) { indent(); out_ << key(name) << ":\n"; ++depth_; }
    void endMap() { if (depth_ > 0) --depth_; }
    void comment(const std::string &text) { indent(); out_ << "# " << text << "\n"; }
    void write(const std::string &name, double v) { indent(); out_ << key(name) << ": " << real(v) << "\n"; }
    void write(const std::string &name, int v) { indent(); out_ << key(name) << ": " << v << "\n"; }
    void writeMatrix(const std::string &name, int rows, int cols, const double *data)
    {
        indent(); out_ << key(name) << ": !!opencv-matrix\n";
        ++depth_;
        indent(); out_ << "rows: " << rows << "\n";
        indent(); out_ << "cols: " << cols << "\n";
        indent(); out_ << "dt: d\n";
        indent(); out_ << "data: [ ";
        size_t col = 3 * (size_t)depth_ + 8;
        for (int i = 0; i < rows * cols; ++i) {
            const std::string s = real(data[i]) + (i + 1 < rows * cols ? "," : "");
            if (col + s.size() + 1 > 78 && i > 0) {
                out_ << "\n";
                for (int k = 0; k < 3 * depth_ + 4; ++k) out_ << ' ';
                col = 3 * (size_t)depth_ + 4;
            }
            out_ << s << " ";
            col += s.size() + 1;
        }
        out_ << "]\n";
        --depth_;
    }
    static std::string real(double v)
    { // icvDoubleToString, OpenCV 2.4 modules/core/src/persistence.cpp
        char buf[64];
        const int iv = (int)std::lrint(v);
        if ((double)iv == v && std::fabs(v) < 1e9) std::snprintf(buf, sizeof(buf), "%d.", iv);
        else if (std::isnan(v)) return ".Nan";
        else if (std::isinf(v)) return v < 0 ? "-.Inf" : ".Inf";
        else std::snprintf(buf, sizeof(buf), "%.16e", v);
        return buf;
    }

private:
    static std::string key(const std::string &k)
    { // names that are not plain identifiers are quoted ("Frame 12")
        bool plain = !k.empty() && (std::isalpha((unsigned char)k[0]) || k[0] == '_');
        for (size_t i = 0; i < k.size() && plain; ++i) plain = std::isalnum((unsigned char)k[i]) || k[i] == '_' || k[i] == '-';
        return plain ? k : "\"" + k + "\"";
    }
    void indent() { for (int k = 0; k < 3 * depth_; ++k) out_ << ' '; }
    std::ofstream out_;
    int depth_;
};

// ------------------------------------------------------------------------------------------------ ImageEKF
// class EKF of the reference (1PointRansacEKF/EKF.h:41-63) with its own constructor arguments, image in.
class ImageEKF {
public:
    // EKF::EKF(pathConfigFile, outputPath)  EKF.cpp:124-146.  detectorThreshold: threshold of this build's corner
    // measure (the reference's detector thresholds live in the FeatureDetector section, which is OpenCV-specific).
    ImageEKF(const char *pathConfigFile, const char *outputPath, int precision = EKF_PRECISION_F64, double detectorThreshold = 1e9)
        : e_(0), steps_(0), outputPath_(outputPath ? outputPath : ""), detectorThreshold_(detectorThreshold)
    {
        std::string err;
        if (!loadConfiguration(pathConfigFile, cam_, par_, run_, &err)) throw std::runtime_error("configuration: " + err);
        EkfEngineConfig cfg = EkfEngineConfig();
        cfg.cam = cam_; cfg.par = par_;
        cfg.max_features = std::max(run_.reserveFeaturesDepth, run_.reserveFeaturesInvDepth);
        cfg.precision = precision; cfg.device = -1;
        const int rc = ekf_engine_create(&cfg, &e_);
        if (rc != EKF_OK) throw std::runtime_error("ekf_engine_create failed (no MI355X visible?)");
        if (!outputPath_.empty()) { // EKF.cpp:129-137: output.yml + log.txt (+ a video this build does not write)
            if (!out_.open(outputPath_ + "output.yml")) throw std::runtime_error("cannot write " + outputPath_ + "output.yml");
            log_.open((outputPath_ + "log.txt").c_str(), std::ios_base::out);
            ekf_timing_enable(e_, 1);
            ekf_keep_step_predictions(e_, 1);
        }
    }
    ~ImageEKF()
    {
        out_.release();
        if (log_.is_open()) log_.close();
        if (e_) ekf_engine_destroy(e_);
    }
    // EKF::init(image)  EKF.cpp:170-237
    void init(const Image &image)
    {
        if (log_.is_open()) { // EKF.cpp:172-180; the new-feature selection of this build draws no random numbers: seed 0
            log_ << "Random Seed: " << 0 << std::endl << std::endl;
            log_ << "~~~~~~~~~~~~ STEP " << steps_ << " ~~~~~~~~~~~~" << std::endl;
        }
        chk(ekf_reset(e_), "ekf_reset");
        chk(ekf_image_upload(e_, image.data.data(), image.width, image.height, image.width * image.channels, image.channels), "ekf_image_upload");
        addNewFeatures(run_.minMatchesPerImage);
        logState();
    }
    // EKF::step(image)  EKF.cpp:242-666
    EkfStepInfo step(const Image &image)
    {
        ++steps_; // EKF.cpp:244: the first step is "Frame 1", map management runs on steps f, 2f, ...
        if (log_.is_open()) {
            log_ << std::endl << std::endl;
            log_ << "~~~~~~~~~~~~ STEP " << steps_ << " ~~~~~~~~~~~~" << std::endl;
        }
        EkfStepInfo info;
        if (out_.isOpened()) ekf_timing_reset(e_);
        chk(ekf_step_image(e_, image.data.data(), image.width, image.height, image.width * image.channels, image.channels, &info), "ekf_step_image");
        if (!outputPath_.empty()) writePredictionImage(image); // EKF.cpp:294-305
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        const int inliers = info.n_inliers + info.n_rescued;
        // EKF.cpp:574-612 (updateMapFeatures already ran inside the step)
        if (run_.mapManagementFrequency > 0 && steps_ % run_.mapManagementFrequency == 0) {
            const int needed = run_.minMatchesPerImage - inliers;
            const int N = ekf_num_features(e_);
            std::vector<uint8_t> drop(N, 0);
            {   // removeBadMapFeatures, MapManagement.cpp:279-308 (float ratio; 0/0 keeps the feature)
                std::vector<uint32_t> tp(N + 1), tm(N + 1);
                chk(ekf_get_map_features(e_, 0, tp.data(), tm.data()), "ekf_get_map_features");
                for (int i = 0; i < N; ++i)
                    if ((float)tm[i] / (float)tp[i] < par_.goodFeatureMatchingPercent) drop[i] = 1;
            }
            // the conditions below are evaluated AFTER removeBadMapFeatures in the reference (EKF.cpp:580-586):
            // map size and covariance rows of the features that are kept
            std::vector<int32_t> type(N + 1);
            chk(ekf_get_feature_layout(e_, type.data(), 0), "ekf_get_feature_layout");
            int kept = 0, rows = 13;
            for (int i = 0; i < N; ++i)
                if (!drop[i]) { ++kept; rows += type[i] == EKF_FEATURE_INVERSE_DEPTH ? 6 : 3; }
            if (needed > 0 && (run_.alwaysRemoveUnseenMapFeatures || (run_.maxMapFeaturesCount > 0 && kept + needed > run_.maxMapFeaturesCount) ||
                               (run_.maxMapSize > 0 && rows + needed * 6 > run_.maxMapSize))) {
                std::vector<int32_t> unseen(N + 1);
                int nu = 0;
                chk(ekf_get_unseen_features(e_, unseen.data(), &nu), "ekf_get_unseen_features");
                for (int i = 0; i < nu; ++i) drop[unseen[i]] = 1;
            }
            std::vector<int32_t> idx;
            for (int i = 0; i < N; ++i)
                if (drop[i]) idx.push_back(i);
            if (!idx.empty()) chk(ekf_remove_features(e_, idx.data(), (int)idx.size()), "ekf_remove_features");
            int conv = -1;
            chk(ekf_convert_inverse_depth_to_depth(e_, &conv), "ekf_convert_inverse_depth_to_depth");
            if (needed > 0) addNewFeatures(needed);
        }
        if (out_.isOpened()) {
            const double mapUs = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            writeFrame(info, mapUs);
        }
        logState();
        return info;
    }
    EkfEngine *engine() { return e_; }
    int steps() const { return steps_; }
    const EkfCamera &camera() const { return cam_; }
    const RunParameters &runParameters() const { return run_; }

private:
    ImageEKF(const ImageEKF &);
    ImageEKF &operator=(const ImageEKF &);
    void chk(int rc, const char *what)
    {
        if (rc != EKF_OK) throw std::runtime_error(std::string(what) + ": " + ekf_last_error(e_));
    }
    // detectNewImageFeatures + addFeaturesToStateAndCovariance (+ templates for matcher mode B)
    void addNewFeatures(int wanted)
    {
        const int N0 = ekf_num_features(e_);
        wanted = std::min(wanted, std::max(run_.reserveFeaturesInvDepth - N0, 0));
        if (wanted <= 0) return;
        std::vector<double> uv(2 * (size_t)wanted);
        int got = 0;
        chk(ekf_detect_new_features(e_, wanted, run_.detectNewFeaturesImageAreasDivideTimes, run_.detectNewFeaturesImageMaskEllipseSize,
                                    detectorThreshold_, uv.data(), &got), "ekf_detect_new_features");
        if (got <= 0) return;
        chk(ekf_add_features(e_, uv.data(), 0, got), "ekf_add_features");
        std::vector<int32_t> idx(got);
        for (int i = 0; i < got; ++i) idx[i] = N0 + i;
        chk(ekf_capture_templates(e_, idx.data(), uv.data(), got), "ekf_capture_templates");
    }
    void logState()
    {   // state.showDetailed(_logFile)  EKF.cpp:233-236, 662-665
        if (!log_.is_open()) return;
        const int N = ekf_num_features(e_);
        double x[13];
        std::vector<double> fp(6 * (size_t)N + 6);
        std::vector<int32_t> type(N + 1);
        std::vector<uint32_t> tp(N + 1), tm(N + 1);
        chk(ekf_get_state(e_, x, fp.data(), 0), "ekf_get_state");
        chk(ekf_get_feature_layout(e_, type.data(), 0), "ekf_get_feature_layout");
        chk(ekf_get_map_features(e_, 0, tp.data(), tm.data()), "ekf_get_map_features");
        showDetailed(log_, x, N, fp.data(), type.data(), tp.data(), tm.data());
    }
    void writePredictionImage(const Image &image)
    {
        int np = 0;
        chk(ekf_get_step_predictions(e_, 0, &np), "ekf_get_step_predictions");
        std::vector<EkfPrediction> preds(np + 1);
        chk(ekf_get_step_predictions(e_, preds.data(), &np), "ekf_get_step_predictions");
        std::vector<int32_t> type(ekf_num_features(e_) + 1);
        chk(ekf_get_feature_layout(e_, type.data(), 0), "ekf_get_feature_layout");
        Image drawn;
        drawPrediction(image, preds.data(), np, type.data(), drawn);
        char name[32];
        std::snprintf(name, sizeof(name), "%05d.png", steps_);
        if (!writePng(outputPath_ + name, drawn)) throw std::runtime_error("cannot write " + outputPath_ + name); // cv::imwrite, EKF.cpp:302
    }
    void writeFrame(const EkfStepInfo &info, double mapManagementUs)
    {
        EkfStageTimes t;
        ekf_timing_get(e_, &t);
        std::ostringstream name;
        name << "Frame " << steps_;
        out_.beginMap(name.str());
        out_.comment(""); out_.comment("Running time (microseconds)"); out_.comment("");
        out_.write("Prediction", 1e3 * t.prediction_ms);
        out_.write("Matching", 1e3 * t.matching_ms);
        out_.write("Ransac", 1e3 * t.ransac_ms);
        out_.write("totalMatches", (int)info.n_matches);
        out_.write("liInliers", (int)info.n_inliers);
        out_.write("UpdateLI", 1e3 * t.update_li_ms);
        out_.write("RescueOutliers", 1e3 * t.rescue_ms);
        out_.write("hiInliers", (int)info.n_rescued);
        out_.write("UpdateHI", 1e3 * t.update_hi_ms);
        out_.write("MapManagement", mapManagementUs);
        out_.comment(""); out_.comment("State and Covariance Estimation"); out_.comment("");
        double x[13], P13[169];
        chk(ekf_get_state(e_, x, 0, 0), "ekf_get_state");
        chk(ekf_get_camera_covariance(e_, P13), "ekf_get_camera_covariance");
        const int N = ekf_num_features(e_);
        std::vector<int32_t> type(N + 1), covpos(N + 1);
        chk(ekf_get_feature_layout(e_, type.data(), covpos.data()), "ekf_get_feature_layout");
        int inv = 0;
        for (int i = 0; i < N; ++i) inv += type[i] == EKF_FEATURE_INVERSE_DEPTH;
        out_.writeMatrix("StateEstimation", 1, 13, x);          // State::write, State.cpp:339-360
        out_.write("MapFeaturesInvDepthCount", inv);
        out_.write("MapFeaturesDepthCount", N - inv);
        out_.writeMatrix("StateCovarianceMatrixEstimation", 13, 13, P13);
        out_.endMap();
    }
    EkfEngine *e_;
    int steps_;
    std::string outputPath_;
    double detectorThreshold_;
    EkfCamera cam_;
    EkfParams par_;
    RunParameters run_;
    OutputWriter out_;
    std::ofstream log_;
};

} // namespace ekf_compat

// ------------------------------------------------------------------------------ class EKF under the reference's own signatures
// EKF(configurationFileName, outputPath), init(image), step(image)  (EKF/EKF.h:44-48) -- declared in ekf_compat.h, defined here
// because they need the configuration loader and the output writer: class EKF drives an ImageEKF and shares its engine.
#include "ekf_compat.h"

#ifndef EKF_COMPAT_PRECISION
#define EKF_COMPAT_PRECISION EKF_PRECISION_F64 // what EKF(configurationFileName, outputPath) creates its engine with (the reference computes in double)
#endif

namespace ekf_compat {
inline Image imageFromMat(const cv::Mat &m)
{
    Image img;
    if (m.empty()) return img;
    img.width = m.cols;
    img.height = m.rows;
    img.channels = m.channels();
    const size_t row = (size_t)m.cols * m.channels();
    img.data.resize(row * m.rows);
    for (int y = 0; y < m.rows; ++y) std::memcpy(&img.data[row * y], m.data + (size_t)y * m.step, row);
    return img;
}
// a view (no copy): `img` must outlive it
inline cv::Mat matFromImage(Image &img) { return cv::Mat(img.height, img.width, img.channels, img.data.data()); }
inline void deleteImageEKF(ImageEKF *p) { delete p; }
} // namespace ekf_compat

inline EKF::EKF(const char *configurationFileName, const char *outputPath) : e_(0), steps_(0), drv_(0), drvDelete_(0)
{
    std::memset(&last_, 0, sizeof(last_));
    drv_ = new ekf_compat::ImageEKF(configurationFileName, outputPath, EKF_COMPAT_PRECISION);
    drvDelete_ = ekf_compat::deleteImageEKF;
    e_ = drv_->engine();
}

inline void EKF::init(const cv::Mat &image)
{
    if (!drv_) throw ekf_compat::Error(EKF_ERR_INVALID_ARG, "EKF::init(image) needs the EKF(configurationFileName, outputPath) constructor");
    drv_->init(ekf_compat::imageFromMat(image));
    refreshLayout();
    ekf_compat::download(e_, state, 0);
    stateCovarianceMatrix.markStale(e_, ekf_state_dim(e_));
}

inline void EKF::step(const cv::Mat &image)
{
    if (!drv_) throw ekf_compat::Error(EKF_ERR_INVALID_ARG, "EKF::step(image) needs the EKF(configurationFileName, outputPath) constructor");
    last_ = drv_->step(ekf_compat::imageFromMat(image));
    ++steps_;
    refreshLayout(); // the reference's public attributes are current after a step: `state` now, the covariance on first access
    ekf_compat::download(e_, state, 0);
    stateCovarianceMatrix.markStale(e_, ekf_state_dim(e_));
}
#endif // EKF_IO_H
