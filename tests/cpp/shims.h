// Minimal stand-ins for cv::Mat and Eigen::Matrix so include/HomographyNet.h can be compiled and exercised in an
// image without OpenCV / Eigen.  TEST SCAFFOLDING for the adapter only (the real build uses the real libraries).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace cv {
struct Mat {
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    std::vector<uint8_t> storage;
    Mat() {}
    Mat(int r, int c, size_t stride) : rows(r), cols(c), step(stride), storage(stride * r) { data = storage.data(); }
};
}  // namespace cv

namespace Eigen {
template <typename T, int R, int C>
struct Matrix {
    T v[R * C];
    T& operator()(int i, int j) { return v[i * C + j]; }
    const T& operator()(int i, int j) const { return v[i * C + j]; }
    T& operator[](int i) { return v[i]; }
    const T& operator[](int i) const { return v[i]; }
    void setZero() { for (auto& x : v) x = T(0); }
    template <typename U> Matrix<U, R, C> cast() const {
        Matrix<U, R, C> o;
        for (int i = 0; i < R * C; i++) o.v[i] = (U)v[i];
        return o;
    }
};
}  // namespace Eigen
