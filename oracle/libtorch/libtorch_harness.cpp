// libtorch_harness.cpp — the reference's CPU inference path as a C++ program (TEST / BASELINE INFRASTRUCTURE, not the product):
// torch::jit::load of a TorchScript file + module.forward on ONE frame pair at a time, exactly the two libtorch calls
// pytorch::HomographyNet makes (reference cuahn_ros/homography_network/src/HomographyNet.cpp:89 `torch::jit::load`, :183-186
// `network.forward(inputs)`), BASELINE.json config 1 and the `cpu_baseline` leg of bench.py.  The TorchScript file is OUR traced
// restatement (oracle/libtorch/trace_restatement.py): the reference's own .pt cannot travel to the GPU box.
//
//   libtorch_harness <model.pt> <inputs.bin> <n_mc> <threads> <seconds> [<outputs.bin>]
// inputs.bin: float32 img1[71680] img2[71680] prior[8] keep_mean_in[n*5120] keep_mean_hid[n*256] keep_unc_in[n*5120] keep_unc_hid[n*256]
// Runs one warm-up forward, then forwards for `seconds` of wall clock (at least 2); prints one JSON line with the timing and the
// outputs of the last forward; outputs.bin (optional) receives mean[8] cov[64] H_part1[9] as float32.
#include <ATen/Parallel.h>
#include <torch/script.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s model.pt inputs.bin n_mc threads seconds [outputs.bin]\n", argv[0]);
        return 2;
    }
    const int n_mc = std::atoi(argv[3]), threads = std::atoi(argv[4]);
    const double seconds = std::atof(argv[5]);
    at::set_num_threads(threads > 0 ? threads : 1);
    torch::NoGradGuard no_grad;
    torch::jit::script::Module module;
    try {
        module = torch::jit::load(argv[1]);                       // HomographyNet.cpp:89
    } catch (const c10::Error& e) {
        std::fprintf(stderr, "error loading the model\n");       // :91-93
        return 1;
    }
    module.eval();
    const size_t n_in = 71680 * 2 + 8 + (size_t)n_mc * (5120 + 256) * 2;
    std::vector<float> buf(n_in);
    std::ifstream f(argv[2], std::ios::binary);
    if (!f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(n_in * sizeof(float)))) {
        std::fprintf(stderr, "inputs.bin: expected %zu floats\n", n_in);
        return 1;
    }
    float* p = buf.data();
    auto take = [&](std::vector<int64_t> shape) {
        int64_t n = 1;
        for (auto d : shape) n *= d;
        torch::Tensor t = torch::from_blob(p, shape, torch::kFloat32).clone();
        p += n;
        return t;
    };
    std::vector<torch::jit::IValue> inputs;
    inputs.push_back(take({224, 320}));
    inputs.push_back(take({224, 320}));
    inputs.push_back(take({8}));
    inputs.push_back(take({n_mc, 5120}));
    inputs.push_back(take({n_mc, 256}));
    inputs.push_back(take({n_mc, 5120}));
    inputs.push_back(take({n_mc, 256}));

    auto out = module.forward(inputs).toTuple();                  // warm-up (HomographyNet.cpp:99-131 runs three)
    int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    double dt = 0.0;
    while (n < 2 || dt < seconds) {
        out = module.forward(inputs).toTuple();                   // :183-186
        n++;
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const torch::Tensor mean = out->elements()[0].toTensor().contiguous();
    const torch::Tensor cov = out->elements()[1].toTensor().contiguous();
    const torch::Tensor h1 = out->elements()[2].toTensor().contiguous();
    std::printf("{\"forwards\": %d, \"seconds\": %.6f, \"ms_per_forward\": %.4f, \"threads\": %d, \"mean\": [", n, dt, 1e3 * dt / n, at::get_num_threads());
    for (int i = 0; i < 8; i++) std::printf("%s%.9g", i ? ", " : "", mean.data_ptr<float>()[i]);
    std::printf("], \"cov_diag\": [");
    for (int i = 0; i < 8; i++) std::printf("%s%.9g", i ? ", " : "", cov.data_ptr<float>()[i * 9]);
    std::printf("]}\n");
    if (argc > 6) {
        std::ofstream o(argv[6], std::ios::binary);
        o.write(reinterpret_cast<const char*>(mean.data_ptr<float>()), 8 * sizeof(float));
        o.write(reinterpret_cast<const char*>(cov.data_ptr<float>()), 64 * sizeof(float));
        o.write(reinterpret_cast<const char*>(h1.data_ptr<float>()), 9 * sizeof(float));
    }
    return 0;
}
