// Drives include/HomographyNet.h the way cuahn::VioManager drives the reference class
// (VioManager.cpp:107,188,236,257-259): construct, feed images, infer with a prior, read mean / Cov.
// usage: adapter_smoke <weights.hnw> <frames.u8 (n x 224 x 320 bytes)> <n_frames> <use_prior 0|1> [<iterative weights.hnw> <num_of_iteration>]
// With an iterative model every frame runs network_inference(prior, it) for it = 0 .. num_of_iteration - 1 (VioManager.cpp:227-275) and prints one
// "RESULT k it ..." line per call: iteration 0 is the main model, iterations > 0 the iterative one (HomographyNet.cpp:183, :211).
#define HNET_ADAPTER_NO_THIRD_PARTY_INCLUDES
#include "shims.h"
#include "../../include/HomographyNet.h"

#include <cstdio>
#include <memory>

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage\n"); return 2; }
    std::string model = argv[1], iter_model = argc > 5 ? argv[5] : "";
    const int n = std::atoi(argv[3]);
    const bool use_prior = std::atoi(argv[4]) != 0;
    const int n_it = argc > 6 ? std::atoi(argv[6]) : 1;
    std::FILE* f = std::fopen(argv[2], "rb");
    if (!f) return 2;
    std::shared_ptr<pytorch::HomographyNet> HNet(new pytorch::HomographyNet(model, iter_model, use_prior, n_it, false));
    Eigen::Matrix<double, 8, 1> prior;
    const double pv[8] = {1.0, -2.0, 0.5, 3.0, -1.5, 0.25, 2.0, -0.75};
    for (int i = 0; i < 8; i++) prior[i] = pv[i];
    for (int k = 0; k < n; k++) {
        cv::Mat img(224, 320, 384);   // row stride larger than the width, like an ROI of a wider image
        for (int r = 0; r < 224; r++)
            if (std::fread(img.data + (size_t)r * img.step, 1, 320, f) != 320) return 2;
        const double t = 100.0 + k;
        HNet->load_current_img(img, t);
        for (int it = 0; it < n_it; it++) {
            HNet->network_inference(prior, it);
            if (HNet->get_latest_inference_time() == t && HNet->img_counter > 1) {
                Eigen::Matrix<double, 8, 1> m = HNet->get_pred_mean();
                Eigen::Matrix<double, 8, 8> c = HNet->get_pred_Cov();
                if (n_it > 1) std::printf("RESULT %d %d", k, it);
                else std::printf("RESULT %d", k);
                for (int i = 0; i < 8; i++) std::printf(" %.9g", m(i, 0));
                for (int i = 0; i < 8; i++)
                    for (int j = 0; j < 8; j++) std::printf(" %.9g", c(i, j));
                std::printf("\n");
            }
        }
    }
    std::fclose(f);
    return 0;
}
