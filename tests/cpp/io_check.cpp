// io_check -- exercises openekfmonoslam_amd/compat/ekf_io.h without a GPU (only the header's host code):
//   io_check config <config.yml>                 -> prints the parsed camera / filter / run parameters
//   io_check png <in.png> <out.png>              -> decodes, prints geometry + byte sum, re-encodes
//   io_check yaml <out.yml>                      -> writes two frames in the output.yml layout
//   io_check draw <out.png>                      -> drawPrediction on a synthetic frame (two predictions)
//   io_check log <out.txt>                       -> State::showDetailed layout
#include <cstdio>
#include <string>

#include "../../openekfmonoslam_amd/compat/ekf_io.h"

int main(int argc, const char *argv[])
{
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    if (mode == "config") {
        EkfCamera cam;
        EkfParams par;
        ekf_compat::RunParameters run;
        std::string err;
        if (!ekf_compat::loadConfiguration(argv[2], cam, par, run, &err)) {
            std::printf("error %s\n", err.c_str());
            return 1;
        }
        std::printf("cam %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", cam.pixelsX, cam.pixelsY, cam.fx,
                    cam.fy, cam.k1, cam.k2, cam.cx, cam.cy, cam.dx, cam.dy, cam.pixelErrorX, cam.pixelErrorY, cam.angularVisionX,
                    cam.angularVisionY);
        std::printf("par %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", par.initInvDepthRho,
                    par.initLinearAccelSD, par.initAngularAccelSD, par.linearAccelSD, par.angularAccelSD, par.inverseDepthRhoSD,
                    par.matchingCompCoefSecondBestVSFirst, par.ransacThresholdPredictDistance, par.ransacAllInliersProbability,
                    par.ransacChi2Threshold, par.goodFeatureMatchingPercent, par.inverseDepthLinearityIndexThreshold);
        std::printf("run %d %d %d %d %d %d %d %.17g %d\n", run.reserveFeaturesDepth, run.reserveFeaturesInvDepth, run.maxMapSize,
                    run.maxMapFeaturesCount, (int)run.alwaysRemoveUnseenMapFeatures, run.mapManagementFrequency,
                    run.detectNewFeaturesImageAreasDivideTimes, run.detectNewFeaturesImageMaskEllipseSize, run.minMatchesPerImage);
        return 0;
    }
    if (mode == "png" && argc >= 4) {
        ekf_compat::Image img;
        std::string err;
        if (!ekf_compat::readPng(argv[2], img, &err)) {
            std::printf("error %s\n", err.c_str());
            return 1;
        }
        unsigned long long sum = 0;
        for (size_t i = 0; i < img.data.size(); ++i) sum += (unsigned long long)img.data[i] * (i % 251 + 1);
        std::printf("png %d %d %d %llu\n", img.width, img.height, img.channels, sum);
        return ekf_compat::writePng(argv[3], img) ? 0 : 1;
    }
    if (mode == "yaml") {
        ekf_compat::OutputWriter w;
        if (!w.open(argv[2])) return 1;
        for (int k = 0; k < 2; ++k) {
            w.beginMap(k == 0 ? "Frame 1" : "Frame 2");
            w.comment("");
            w.comment("Running time (microseconds)");
            w.write("Prediction", 123.456 + k);
            w.write("totalMatches", 57 + k);
            w.write("UpdateLI", 2000.0);
            double x[13], P[169];
            for (int i = 0; i < 13; ++i) x[i] = (i + 1) / 7.0 * (k ? -1 : 1);
            for (int i = 0; i < 169; ++i) P[i] = i % 14 == 0 ? 1e-6 * (i + 1) : 0.0;
            w.writeMatrix("StateEstimation", 1, 13, x);
            w.write("MapFeaturesInvDepthCount", 40);
            w.writeMatrix("StateCovarianceMatrixEstimation", 13, 13, P);
            w.endMap();
        }
        w.release();
        return 0;
    }
    if (mode == "draw") {
        ekf_compat::Image img, out;
        img.width = 96; img.height = 64; img.channels = 1;
        img.data.assign((size_t)96 * 64, 40);
        EkfPrediction p[2];
        p[0].featureIndex = 0; p[0].imagePos[0] = 30.7; p[0].imagePos[1] = 20.2;
        p[0].covarianceMatrix[0] = 4.0; p[0].covarianceMatrix[1] = 0.0; p[0].covarianceMatrix[2] = 0.0; p[0].covarianceMatrix[3] = 1.0;
        p[1].featureIndex = 1; p[1].imagePos[0] = 70.0; p[1].imagePos[1] = 40.0;
        p[1].covarianceMatrix[0] = 1.0; p[1].covarianceMatrix[1] = 0.0; p[1].covarianceMatrix[2] = 0.0; p[1].covarianceMatrix[3] = 1.0;
        const int32_t type[2] = {EKF_FEATURE_INVERSE_DEPTH, EKF_FEATURE_DEPTH};
        ekf_compat::drawPrediction(img, p, 2, type, out);
        return ekf_compat::writePng(argv[2], out) ? 0 : 1;
    }
    if (mode == "log") {
        std::ofstream f(argv[2]);
        const double x[13] = {0.5, -0.25, 1.0, 1, 0, 0, 0, 0.01, 0, 0.002, 0, 0.002, 0};
        const double fp[12] = {0, 0, 0, 0.1, -0.2, 0.5, 1.5, 2.5, 3.5, 0, 0, 0};
        const int32_t type[2] = {EKF_FEATURE_INVERSE_DEPTH, EKF_FEATURE_DEPTH};
        const uint32_t tp[2] = {7, 3}, tm[2] = {5, 3};
        ekf_compat::showDetailed(f, x, 2, fp, type, tp, tm);
        return 0;
    }
    return 2;
}
