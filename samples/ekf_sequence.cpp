// ekf_sequence -- the reference's sample program (kalmanFilter/samples/EKF/main.cpp:45-160) on the MI355X engine:
//     ekf_sequence config.yml imgdir/ [outdir/ [first [last [detector_threshold [f32]]]]]
// reads imgdir/%05d.png from `first` (default 0; the reference hard-codes 90..6550) until `last` or the first missing
// file, initialises the filter on the first frame, steps on the rest and, when outdir is given, writes
// outdir/output.yml in the reference's layout.  Matcher mode B (NCC templates), so no OpenCV is needed.
//
//   g++ -std=c++11 -O2 samples/ekf_sequence.cpp -o ekf_sequence -Lopenekfmonoslam_amd -lekf_engine -lz
//   (plus -Wl,-rpath,$PWD/openekfmonoslam_amd -Wl,-rpath,/opt/rocm/lib)
#include <cstdio>
#include <cstdlib>

#include "../openekfmonoslam_amd/compat/ekf_io.h"

int main(int argc, const char *argv[])
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s config.yml imgdir/ [outdir/ [first [last [detector_threshold [f32]]]]]\n", argv[0]);
        return 2;
    }
    const std::string outputPath = argc > 3 ? argv[3] : "";
    const int first = argc > 4 ? std::atoi(argv[4]) : 0, last = argc > 5 ? std::atoi(argv[5]) : 99999;
    const double threshold = argc > 6 ? std::atof(argv[6]) : 1e9;
    const int precision = (argc > 7 && std::string(argv[7]) == "f32") ? EKF_PRECISION_F32 : EKF_PRECISION_F64;
    try {
        ekf_compat::FileSequenceImageGenerator generator(argv[2], "", "png", first, last);
        generator.init();
        ekf_compat::Image image = generator.getNextImage();
        if (image.empty()) {
            std::printf("No se puede iniciar Kalman Filter dado que no hay imagenes disponibles.\n");
            return 0;
        }
        if (precision == EKF_PRECISION_F64 && threshold == 1e9) {
            // the reference's own three lines (samples/EKF/main.cpp:76-131): EKF(config, outputPath), init(image), step(image)
            EKF extendedKalmanFilter(argv[1], outputPath.c_str());
            extendedKalmanFilter.init(ekf_compat::matFromImage(image));
            std::printf("init: %d features\n", (int)extendedKalmanFilter.state.mapFeatures.size());
            image = generator.getNextImage();
            int steps = 0;
            while (!image.empty()) {
                extendedKalmanFilter.step(ekf_compat::matFromImage(image));
                const EkfStepInfo &info = extendedKalmanFilter.lastStepInfo();
                const double *x = extendedKalmanFilter.state.position;
                std::printf("step %d: predicted %d matches %d li %d hi %d features %d  r = %.6f %.6f %.6f\n", ++steps, info.n_predicted,
                            info.n_matches, info.n_inliers, info.n_rescued, (int)extendedKalmanFilter.state.mapFeatures.size(), x[0], x[1], x[2]);
                image = generator.getNextImage();
            }
            return 0;
        }
        // (a detector threshold or the fp32 configuration asked for: the driver class with its extra constructor arguments)
        ekf_compat::ImageEKF extendedKalmanFilter(argv[1], outputPath.c_str(), precision, threshold);
        extendedKalmanFilter.init(image);
        std::printf("init: %d features\n", ekf_num_features(extendedKalmanFilter.engine()));
        image = generator.getNextImage();
        while (!image.empty()) {
            const EkfStepInfo info = extendedKalmanFilter.step(image);
            double x[13];
            ekf_get_state(extendedKalmanFilter.engine(), x, 0, 0);
            std::printf("step %d: predicted %d matches %d li %d hi %d features %d  r = %.6f %.6f %.6f\n", extendedKalmanFilter.steps(),
                        info.n_predicted, info.n_matches, info.n_inliers, info.n_rescued, ekf_num_features(extendedKalmanFilter.engine()),
                        x[0], x[1], x[2]);
            image = generator.getNextImage();
        }
    } catch (const std::exception &ex) {
        std::fprintf(stderr, "ekf_sequence: %s\n", ex.what());
        return 1;
    }
    return 0;
}
