// tests/cpp/iekf_demo.cpp — the reference's per-frame filter step around the network (VioManager.cpp:188,227-275)
// with include/HomographyNet.h (the drop-in class) and include/hnet_ekf.h (the update that consumes its output):
// for every frame: load_current_img, then the iterated EKF loop (network_inference with the state's offsets x 159.5
// as prior, update, offsets reset).  Prints the state after each frame for tests/test_adapter_cpp.py.
// usage: iekf_demo <weights.hnw> <frames.u8> <n_frames> <max_IEKF_iteration>
#define HNET_ADAPTER_NO_THIRD_PARTY_INCLUDES
#include "shims.h"
#include "../../include/HomographyNet.h"
#include "../../include/hnet_ekf.h"

#include <cstdio>
#include <memory>

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage\n"); return 2; }
    std::string model = argv[1], iter_model = "";
    const int n = std::atoi(argv[3]), max_it = std::atoi(argv[4]);
    std::FILE* f = std::fopen(argv[2], "rb");
    if (!f) return 2;
    std::shared_ptr<pytorch::HomographyNet> HNet(new pytorch::HomographyNet(model, iter_model, true, max_it, false));
    hnet_ekf::State st = {};
    st.q[0] = 1.0;
    for (int i = 0; i < hnet_ekf::NS; i++) st.cov[i * hnet_ekf::NS + i] = i < 15 ? 1e-4 : 0.0;
    Eigen::Matrix<double, 8, 1> prior;
    for (int k = 0; k < n; k++) {
        cv::Mat img(224, 320, 320);
        if (std::fread(img.data, 1, 224 * 320, f) != 224 * 320) return 2;
        HNet->load_current_img(img, 10.0 + k);
        if (HNet->img_counter < 2) continue;
        // stand-in for the IMU propagation (not part of this step): offsets drift, their covariance grows
        for (int c = 0; c < 4; c++) {
            st.offset[c][0] += 0.004 * (c + 1);
            st.offset[c][1] -= 0.003 * (c + 1);
            for (int d = 0; d < 3; d++) {
                const int o = 15 + 3 * c + d;
                st.cov[o * hnet_ekf::NS + o] += 2.5e-3;
                st.cov[d * hnet_ekf::NS + o] = st.cov[o * hnet_ekf::NS + d] = 2e-5;              // position <-> offset
                st.cov[(6 + d) * hnet_ekf::NS + o] = st.cov[o * hnet_ekf::NS + 6 + d] = -1e-5;    // velocity <-> offset
            }
        }
        const int done = hnet_ekf::iterated_update(st, *HNet, max_it, 10.0, prior);
        std::printf("STATE %d %d", k, done);
        for (int i = 0; i < 3; i++) std::printf(" %.17g", st.p[i]);
        for (int i = 0; i < 4; i++) std::printf(" %.17g", st.q[i]);
        for (int i = 0; i < 3; i++) std::printf(" %.17g", st.v[i]);
        for (int i = 0; i < 15; i++) std::printf(" %.17g", st.cov[i * hnet_ekf::NS + i]);
        std::printf("\n");
    }
    std::fclose(f);
    return 0;
}
