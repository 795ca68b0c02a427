// tests/cpp/iekf_demo.cpp — the reference's per-frame filter step around the network (VioManager.cpp:188,227-275)
// with include/HomographyNet.h (the drop-in class) and include/hnet_ekf.h (the update that consumes its output):
// for every frame: load_current_img, IMU propagation of mean and covariance (hnet_ekf::propagate = Propagator::predict_and_compute
// + StateHelper::propagate_Cov over 16 IMU intervals of 2 ms with a constant synthetic rate / specific force), then the iterated
// EKF loop (network_inference with the state's offsets x 159.5 as prior, update gated as VioManager.cpp:257, offsets reset).
// Prints the state after each frame for tests/test_adapter_cpp.py.
// usage: iekf_demo <weights.hnw> <frames.u8> <n_frames> <max_IEKF_iteration> [timing.csv]
// With a fifth argument the per-frame timing file of VioManager.cpp:98,304-311 is written (include/hnet_timing_csv.h).
#define HNET_ADAPTER_NO_THIRD_PARTY_INCLUDES
#include "shims.h"
#include "../../include/HomographyNet.h"
#include "../../include/hnet_ekf.h"
#include "../../include/hnet_timing_csv.h"

#include <chrono>

#include <cstdio>
#include <memory>

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage\n"); return 2; }
    std::string model = argv[1], iter_model = argv[1];      // (the reference's launch files name the same traced file twice unless an iterative variant was traced)
    const int n = std::atoi(argv[3]), max_it = std::atoi(argv[4]);
    std::FILE* f = std::fopen(argv[2], "rb");
    if (!f) return 2;
    std::shared_ptr<pytorch::HomographyNet> HNet(new pytorch::HomographyNet(model, iter_model, true, max_it, false));
    hnet_ekf::State st = {};
    st.q[0] = 1.0;
    st.p[2] = 1.5;                                   // height above the ground (the filter's dc)
    st.v[0] = 0.4; st.v[1] = -0.2;
    for (int i = 0; i < hnet_ekf::NS; i++) st.cov[i * hnet_ekf::NS + i] = i < 15 ? 1e-4 : 0.0;
    Eigen::Matrix<double, 8, 1> prior;
    hnet_csv::TimingCsv csv;
    if (argc > 5 && !csv.open(argv[5])) return 2;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (int k = 0; k < n; k++) {
        cv::Mat img(224, 320, 320);
        if (std::fread(img.data, 1, 224 * 320, f) != 224 * 320) return 2;
        const auto rT1 = now();
        HNet->load_current_img(img, 10.0 + k);
        const auto rT2 = now();
        if (HNet->img_counter < 2) continue;
        // IMU propagation between the two frames: camera 45 degrees down (uzhfpv.launch:84-91), 1.5 m above the ground
        {
            const hnet_ekf::Extrinsics ex = {{-0.027256691772188965, -0.9996260641688061, 0.0021919370477445077,
                                              -0.7139206120417471, 0.017931469899155242, -0.6999970157716363,
                                              0.6996959571525168, -0.020644471939022302, -0.714142404092339},
                                             {0.0070507, 0.0240435, 0.0057731}};
            const double w_hat[3] = {0.02, -0.03, 0.05}, a_hat[3] = {0.1, -0.05, 9.81};
            double qn[hnet_ekf::NW];
            hnet_ekf::noise_q_diag(0.00559017, 0.01118034, 8.94427e-04, 0.04472136, qn);      // uzhfpv.launch:68-71
            for (int i = 0; i < 16; i++) hnet_ekf::propagate(st, ex, 0.002, w_hat, a_hat, qn);
        }
        const auto rT3 = now();
        const double nn_before = HNet->total_host_ms();
        const int done = hnet_ekf::iterated_update(st, *HNet, max_it, 10.0, prior, 10.0 + k);
        const auto rT5 = now();
        if (csv.is_open()) {       // network inference = the host time of this frame's network_inference calls, the rest of the loop = EKF update
            const double nn_ms = HNet->total_host_ms() - nn_before;     // measured inside [rT3, rT5]: never more than the total
            csv.append(10.0 + k - 0.0148489 /* calib_camimu_dt, uzhfpv.launch:43 */, ms(rT1, rT2), ms(rT2, rT3), nn_ms,
                       std::max(0.0, ms(rT3, rT5) - nn_ms), ms(rT1, rT5));
        }
        std::printf("STATE %d %d", k, done);
        for (int i = 0; i < 3; i++) std::printf(" %.17g", st.p[i]);
        for (int i = 0; i < 4; i++) std::printf(" %.17g", st.q[i]);
        for (int i = 0; i < 3; i++) std::printf(" %.17g", st.v[i]);
        for (int i = 0; i < 15; i++) std::printf(" %.17g", st.cov[i * hnet_ekf::NS + i]);
        std::printf(" %.17g %.17g", st.cov[0 * hnet_ekf::NS + 7], st.cov[4 * hnet_ekf::NS + 13]);     // two off-diagonal entries
        std::printf("\n");
    }
    std::fclose(f);
    return 0;
}
