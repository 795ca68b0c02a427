// per-frame wall clock of the drop-in class (load_current_img + network_inference), the way VioManager drives it
// usage: adapter_latency <weights.hnw> <n_frames> <use_prior 0|1>
#define HNET_ADAPTER_NO_THIRD_PARTY_INCLUDES
#include "shims.h"
#include "../../include/HomographyNet.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <memory>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::string model = argv[1], iter_model = "";
    const int n = std::atoi(argv[2]);
    const bool use_prior = std::atoi(argv[3]) != 0;
    if (!std::freopen("/dev/null", "w", stdout)) return 2;            // the class prints like the reference does
    std::shared_ptr<pytorch::HomographyNet> HNet(new pytorch::HomographyNet(model, iter_model, use_prior, 1, false));
    Eigen::Matrix<double, 8, 1> prior;
    for (int i = 0; i < 8; i++) prior[i] = 0.5 * (i - 3);
    cv::Mat img(224, 320, 320);
    std::vector<double> ms;
    uint32_t s = 1;
    for (int k = 0; k < n; k++) {
        for (int i = 0; i < 224 * 320; i++) { s = s * 1664525u + 1013904223u; img.data[i] = (uint8_t)(s >> 24); }
        const auto t0 = std::chrono::steady_clock::now();
        HNet->load_current_img(img, 1.0 + k);
        HNet->network_inference(prior, 0);
        volatile double sink = HNet->get_pred_mean()(0, 0) + HNet->get_pred_Cov()(7, 7);
        (void)sink;
        const auto t1 = std::chrono::steady_clock::now();
        if (k >= 20) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
    std::sort(ms.begin(), ms.end());
    std::fprintf(stderr, "LATENCY p50 %.4f p95 %.4f ms over %zu frames\n", ms[ms.size() / 2], ms[ms.size() * 95 / 100], ms.size());
    return 0;
}
