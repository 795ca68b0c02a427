// HomographyNet.h — header-only C++ adapter with the class surface of the reference runtime
// `pytorch::HomographyNet` (reference cuahn_ros/homography_network/src/HomographyNet.h:23-67), implemented on
// the C ABI of hnet.h (libhnet_hip.so, hand-written HIP kernels for gfx950) instead of libtorch.
//
// Drop-in use: put this header in place of the reference's HomographyNet.h, drop HomographyNet.cpp from
// `network_lib`, link libhnet_hip.so.  The two call sites in cuahn::VioManager
// (VioManager.cpp:107,188,236,257-259,288) compile unchanged:
//
//     HNet = std::shared_ptr<pytorch::HomographyNet>(new pytorch::HomographyNet(model_path, iter_path,
//                                                     use_prior, max_IEKF_iteration, show_img));
//     HNet->load_current_img(standard_img, t);                       // cv::Mat 224x320 CV_8UC1
//     HNet->network_inference(propagated_4pt_offset_pixel, iteration);
//     if (HNet->get_latest_inference_time() == t && HNet->img_counter > 10) { HNet->get_pred_mean(); HNet->get_pred_Cov(); }
//
// Differences to the reference, all at construction:
//   * `network_model_path` names an HNETW001 weight blob (cuahn_vio_amd/weights.py) instead of a TorchScript
//     .pt.  As in the reference, "_showError" in the file name selects the variant that also emits the
//     photometric error map (HomographyNet.cpp:96-100).
//   * what the reference freezes into the traced file (blocks_to_run, MC-dropout N and rate, the error-map twin: trace_model.py:16,36-46) is read from
//     the HNETW001 file's `hnet.variant` record (python -m cuahn_vio_amd.weights --variant ...); a record-less file gives the reference's launch values
//     (3 blocks with prior, N 16, p 0.05).  HNET_BLOCKS_TO_RUN / HNET_MC_SAMPLES / HNET_DROPOUT_P / HNET_ITER_BLOCKS_TO_RUN are explicit OVERRIDES for
//     experiments; HNET_MC_SEED (default 0), HNET_DEVICE (default 0) and HNET_PRECISION (hnet.h) are run-time settings, not properties of the file.
//     The constructor's `use_prior` must agree with the record (a `full` file opened with use_prior = true throws, like the Python mirror).
//   * the IEKF "iterative" model (HomographyNet.cpp:20-24,104-124): with num_of_iteration > 1 a SECOND context is created from
//     `network_model_iterative_path` (its own "_showError" sniff - as in the reference the flag of the file loaded last wins, :117-121 -
//     and its own HNET_ITER_BLOCKS_TO_RUN, default HNET_BLOCKS_TO_RUN), warmed up like the first (:49-56), attached to the main
//     context's frames (hnet_attach_images) and used for every call with iteration > 0 (:209-219).
//   * a failed load prints the reference's "error loading the model !!!" to stderr and throws std::runtime_error out of the constructor.  The reference
//     prints and carries on (HomographyNet.cpp:91-93) - into its own warm-up forward on an empty module (:28-45), which throws c10::Error out of the same
//     constructor: either way `new pytorch::HomographyNet(...)` at VioManager.cpp:107 (no try block there) ends the node before the first frame; here with
//     the status text of the C ABI in what().  INTEGRATION.md section 5.
//
// OpenCV / Eigen types appear only in this header; the shared library has none of them in its ABI.
#ifndef PYTORCH_HNet_H
#define PYTORCH_HNet_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#ifndef HNET_ADAPTER_NO_THIRD_PARTY_INCLUDES
#include <Eigen/Eigen>
#include <opencv2/core/core.hpp>
#include <opencv2/highgui/highgui.hpp>
#endif

#include "hnet.h"

#define HNET_BLUE "\033[34m"
#define HNET_RESET "\033[0m"

namespace pytorch {

class HomographyNet {
public:
    HomographyNet(std::string& network_model_path, std::string& network_model_iterative_path, bool use_prior,
                  int num_of_iteration, bool show_imgs) {
        use_prior_4pt_offset = use_prior;
        iteration = num_of_iteration > 1;
        cv_imshow = show_imgs;
        const bool main_err = network_model_path.find("_showError") != std::string::npos;       // HomographyNet.cpp:96-100
        const bool iter_err = network_model_iterative_path.find("_showError") != std::string::npos;   // :117-121
        // The model VARIANT - blocks_to_run, MC-dropout N and rate, the error-map twin - is frozen into the traced .pt the reference loads (trace_model.py:16,36-46;
        // HomographyNet.cpp:81-124 only names a file).  Here it comes from the HNETW001 file's `hnet.variant` record (python -m cuahn_vio_amd.weights --variant ...):
        // the fields are left at HNET_FROM_FILE.  HNET_BLOCKS_TO_RUN / HNET_MC_SAMPLES / HNET_DROPOUT_P / HNET_ITER_BLOCKS_TO_RUN are explicit OVERRIDES for
        // experiments (a record-less blob + no override = the reference's launch values); what is in effect is printed below.
        hnet_config cfg;
        hnet_default_config(&cfg);
        cfg.device_id = env_int("HNET_DEVICE", 0);
        // the constructor argument decides what network_inference passes (HomographyNet.cpp:160-172); with use_prior the file's record is read and must agree
        cfg.use_prior = use_prior ? HNET_FROM_FILE : 0;
        cfg.blocks_to_run = env_int("HNET_BLOCKS_TO_RUN", HNET_FROM_FILE);
        cfg.mc_samples = env_int("HNET_MC_SAMPLES", HNET_FROM_FILE);
        cfg.dropout_p = (float)env_double("HNET_DROPOUT_P", -1.0);
        cfg.mc_seed = (uint64_t)env_double("HNET_MC_SEED", 0.0);      // (run-time state of PyTorch's generator in the reference, not a property of the file)
        cfg.precision = env_int("HNET_PRECISION", cfg.precision);      // HNET_PREC_F16X2 (3, default), HNET_PREC_BF16X3 (2), HNET_PREC_FP32 (0), HNET_PREC_BF16 (1)
        // the "_showError" file name forces the map like the reference's sniff (:96-100); otherwise the file's record decides.
        // With an iterative model the main model's map is never read (:199)
        cfg.emit_error_map = iteration ? 0 : (main_err ? 1 : HNET_FROM_FILE);
        cfg.max_batch = 1;
        std::printf("Loading the Network Model (HNETW001 weights) from %s ...\n", network_model_path.c_str());
        int rc = hnet_create(&cfg, network_model_path.c_str(), &ctx_);   // also runs the warm-up forward (:28-45)
        if (rc != HNET_OK) {
            std::fprintf(stderr, "error loading the model !!!\n%s\n", network_model_path.c_str());      // the reference's two lines (:92,95)
            throw std::runtime_error(std::string("error loading the model !!! (") + hnet_status_string(rc) + ")");
        }
        require_prior_agrees(ctx_, use_prior, "main model");
        print_variant("main model", ctx_);
        hnet_timing t;
        hnet_last_timing(ctx_, &t);
        std::printf(HNET_BLUE "[TIME]: %.4f milliseconds for the first network inference\n" HNET_RESET, t.host_ms);
        hnet_config used;
        hnet_get_config(ctx_, &used);
        bool any_err = used.emit_error_map != 0;
        if (iteration) {                                               // :20-24, warm-up :49-56
            hnet_config ci = cfg;
            ci.blocks_to_run = env_int("HNET_ITER_BLOCKS_TO_RUN", HNET_FROM_FILE);      // a separate traced file: its own variant
            ci.emit_error_map = iter_err ? 1 : HNET_FROM_FILE;
            std::printf("Loading the Network Model for IEKF (HNETW001 weights) from %s ...\n", network_model_iterative_path.c_str());
            rc = hnet_create(&ci, network_model_iterative_path.c_str(), &ctx_iter_);
            if (rc == HNET_OK) rc = hnet_attach_images(ctx_iter_, ctx_);
            if (rc != HNET_OK) {
                std::fprintf(stderr, "error loading the model !!!\n%s\n", network_model_iterative_path.c_str());
                hnet_destroy(ctx_iter_);
                hnet_destroy(ctx_);
                ctx_iter_ = ctx_ = nullptr;
                throw std::runtime_error(std::string("error loading the model !!! (iterative: ") + hnet_status_string(rc) + ")");
            }
            require_prior_agrees(ctx_iter_, use_prior, "iterative model");
            print_variant("iterative model", ctx_iter_);
            hnet_get_config(ctx_iter_, &used);
            any_err = used.emit_error_map != 0;
            std::printf("IEKF! Load the Network for Iteration!\n");
        }
        show_phtometric_error = any_err;                               // one member in the reference: the file loaded last decides
        if (show_phtometric_error) err_map_.resize(HNET_IMG_ROWS * HNET_IMG_COLS);
        _pred_mean.setZero();
        _pred_Cov.setZero();
#ifndef HNET_ADAPTER_NO_THIRD_PARTY_INCLUDES
        if (cv_imshow) {
            cv::namedWindow("Image", cv::WINDOW_KEEPRATIO);
            cv::namedWindow("Photometric Error", cv::WINDOW_KEEPRATIO);
        }
#endif
    }

    ~HomographyNet() {
        hnet_destroy(ctx_iter_);
        hnet_destroy(ctx_);
        std::printf("HomographyNet Object is being deleted! End of this run ...\n");
    }
    HomographyNet(const HomographyNet&) = delete;
    HomographyNet& operator=(const HomographyNet&) = delete;

    // HomographyNet.cpp:127-151
    void load_current_img(const cv::Mat& img, const double& time_stamp) {
#ifndef HNET_ADAPTER_NO_THIRD_PARTY_INCLUDES
        if (cv_imshow) { cv::imshow("Image", img); cv::waitKey(1); }
#endif
        if (img_counter == 0) std::printf("First Image Comes into the Network Object!\n");
        const int rc = hnet_push_image(ctx_, img.data, img.rows, img.cols, (int)img.step, time_stamp);
        if (rc != HNET_OK) { std::fprintf(stderr, "load_current_img: %s (%s)\n", hnet_status_string(rc), hnet_last_error(ctx_)); return; }
        img_counter = hnet_image_count(ctx_);
    }

    Eigen::Matrix<double, 8, 1> get_pred_mean() { return _pred_mean.template cast<double>(); }
    Eigen::Matrix<double, 8, 8> get_pred_Cov() { return _pred_Cov.template cast<double>(); }
    double get_latest_inference_time() { return hnet_latest_time(ctx_); }
    // extension: wall time of the last network_inference call (ms), for the timing CSV of VioManager.cpp:304-311
    double last_host_ms() { hnet_timing t; hnet_last_timing(last_net_ ? last_net_ : ctx_, &t); return t.host_ms; }
    // extension: the same, summed over every network_inference call so far (a frame's share = the difference across its IEKF iterations)
    double total_host_ms() const { return host_ms_total_; }

    // HomographyNet.cpp:153-252
    void network_inference(Eigen::Matrix<double, 8, 1>& prior_4pt_offset_vec, int num_of_inference) {
        if (img_counter < 2) { std::printf("HNet cannot inference! Only has one image!\n"); return; }   // :155-158
        double prior[8];
        for (int i = 0; i < 8; i++) prior[i] = prior_4pt_offset_vec[i];
        float mean[8], cov[64];
        const bool want_err = cv_imshow && show_phtometric_error;
        hnet_ctx* net = (iteration && num_of_inference > 0) ? ctx_iter_ : ctx_;       // HomographyNet_model / HomographyNet_model_iterative (:183, :211)
        const bool net_err = show_phtometric_error && (net == ctx_iter_ || !iteration);
        last_net_ = net;
        const int rc = hnet_infer(net, use_prior_4pt_offset ? prior : nullptr, num_of_inference, mean, cov, net_err ? err_map_.data() : nullptr);
        if (rc != HNET_OK) { std::fprintf(stderr, "network_inference: %s (%s)\n", hnet_status_string(rc), hnet_last_error(net)); return; }
        { hnet_timing t; if (hnet_last_timing(net, &t) == HNET_OK) host_ms_total_ += t.host_ms; }
        for (int i = 0; i < 8; i++) {
            _pred_mean(i, 0) = mean[i];
            for (int j = 0; j < 8; j++) _pred_Cov(i, j) = cov[i * 8 + j];   // symmetric: the reference's column-major Map of row-major data is the same matrix
        }
#ifndef HNET_ADAPTER_NO_THIRD_PARTY_INCLUDES
        if (want_err && !(num_of_inference == 0 && iteration)) {            // :199, :222
            cv::Mat resultImg(HNET_IMG_ROWS, HNET_IMG_COLS, CV_8UC1, err_map_.data());
            cv::imshow("Photometric Error", resultImg);
            cv::waitKey(1);
        }
#else
        (void)want_err;
#endif
        if (num_of_inference == 0) {                                        // :245-251
            hnet_timing t;
            hnet_last_timing(ctx_, &t);
            if (t.n_main_inferences > 100)                                  // inference_counting: iteration-0 calls only (:189)
                std::printf(HNET_BLUE "[TIME]: %.3f (avg. = %.3f) milliseconds for pure network inference\n" HNET_RESET,
                            t.device_ms, t.sum_device_ms_after_100 / (double)(t.n_main_inferences - 100));
        }
    }

    const std::vector<uint8_t>& last_error_map() const { return err_map_; }   // extension: the u8 map the reference only displays

    // Extension (not in the reference class): undistortion + resize on the GPU instead of CamBase::undistort_and_resize_img
    // on the CPU (CamBase.h:165-186, VioManager.cpp:184).  set_camera = initialize_undist_map[_fisheye]; load_raw_img takes the raw
    // camera frame.  Parity with cv::remap is unpinned (hnet.h).
    bool set_camera(bool fisheye, int raw_rows, int raw_cols, const double k[4], const double d[4]) {
        hnet_camera cam;
        cam.fisheye = fisheye ? 1 : 0; cam.raw_rows = raw_rows; cam.raw_cols = raw_cols;
        for (int i = 0; i < 4; i++) { cam.k[i] = k[i]; cam.d[i] = d[i]; }
        return hnet_set_camera(ctx_, &cam) == HNET_OK;
    }
    void load_raw_img(const cv::Mat& raw, const double& time_stamp) {
        if (img_counter == 0) std::printf("First Image Comes into the Network Object!\n");
        const int rc = hnet_push_raw_image(ctx_, raw.data, raw.rows, raw.cols, (int)raw.step, time_stamp);
        if (rc != HNET_OK) { std::fprintf(stderr, "load_raw_img: %s (%s)\n", hnet_status_string(rc), hnet_last_error(ctx_)); return; }
        img_counter = hnet_image_count(ctx_);
    }

    int img_counter = 0;   // public in the reference (HomographyNet.h:33), read by VioManager.cpp:257,288

private:
    static void print_variant(const char* what, hnet_ctx* c) {
        hnet_config u;
        if (hnet_get_config(c, &u) != HNET_OK) return;
        std::printf("%s: %s, blocks_to_run %d, MC-dropout N = %d, p = %g, error map %s\n", what, u.use_prior ? "EKF prior" : "no prior (4 blocks)",
                    u.use_prior ? u.blocks_to_run : 3, u.mc_samples, (double)u.dropout_p, u.emit_error_map ? "on" : "off");
    }
    // the file's record against the constructor argument (the Python mirror raises ValueError for the same mismatch): a `full` file opened with use_prior = true
    // would otherwise run as prior-3 on a prior the model variant was not meant for.  The contexts are released before the throw (no destructor runs).
    void require_prior_agrees(hnet_ctx* c, bool use_prior, const char* what) {
        hnet_config u;
        if (hnet_get_config(c, &u) == HNET_OK && (u.use_prior != 0) == use_prior) return;
        hnet_destroy(ctx_iter_);
        hnet_destroy(ctx_);
        ctx_iter_ = ctx_ = nullptr;
        throw std::runtime_error(std::string(what) + ": use_prior disagrees with the variant recorded in the weight file");
    }
    static int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
    static double env_double(const char* name, double dflt) { const char* v = std::getenv(name); return v ? std::atof(v) : dflt; }

    hnet_ctx* ctx_ = nullptr;
    hnet_ctx* ctx_iter_ = nullptr;      // the IEKF's second model (num_of_iteration > 1)
    hnet_ctx* last_net_ = nullptr;      // the context of the last network_inference call
    double host_ms_total_ = 0.0;        // wall time of all network_inference calls (total_host_ms)
    bool cv_imshow = false;
    bool use_prior_4pt_offset = false;
    bool show_phtometric_error = false;
    bool iteration = false;
    std::vector<uint8_t> err_map_;
    Eigen::Matrix<float, 8, 1> _pred_mean;
    Eigen::Matrix<float, 8, 8> _pred_Cov;
};

}  // namespace pytorch

#endif  // PYTORCH_HNet_H
