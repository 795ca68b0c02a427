/* hnet_timing_csv.h — the per-frame timing file of the reference, dependency free.
 *
 * VioManager writes one CSV row per processed frame when `record_timing_information` is set
 * (cuahn/src/core/VioManager.cpp:85-99 opens the file — an existing one is deleted first — and writes the header;
 * :299-311 appends `timestamp_inI, load image, propagation, network inference, EKF update, total` with
 * std::fixed / setprecision(15) for the time stamp and setprecision(5) for the five millisecond figures).
 * ov_eval's timing tools parse exactly that (ov_eval/src/utils/Loader.cpp:236-300: a '#' header line with
 * comma-separated category names, then comma-separated numbers).  The reference ships one such file,
 * ov_data/uzh_fpv/traj_timing.txt; tests/test_timing_csv.py re-emits its first rows through this writer and compares
 * the bytes.  A replacement stack that wants `ov_eval timing_*` to keep working fills the five figures from its own
 * clocks (network inference = hnet_last_timing().host_ms of the frame's network_inference calls). */
#ifndef HNET_TIMING_CSV_H
#define HNET_TIMING_CSV_H

#include <cerrno>
#include <cstdio>
#include <string>
#include <sys/stat.h>

namespace hnet_csv {

constexpr const char* kHeader = "# timestamp, loading image, state propagation, network inference, EKF update, total time";

class TimingCsv {
  public:
    TimingCsv() : f_(nullptr) {}
    ~TimingCsv() { close(); }
    TimingCsv(const TimingCsv&) = delete;
    TimingCsv& operator=(const TimingCsv&) = delete;

    /* VioManager.cpp:85-99: delete an old file, open in append mode, write the header */
    bool open(const char* path) {
        close();
        make_parent_dirs(path);                /* boost::filesystem::create_directories(p.parent_path()), VioManager.cpp:92-93 */
        std::remove(path);
        f_ = std::fopen(path, "a");
        if (!f_) return false;
        std::fprintf(f_, "%s\n", kHeader);
        return true;
    }
    bool is_open() const { return f_ != nullptr; }

    /* VioManager.cpp:304-311; all durations in milliseconds, timestamp_inI = state timestamp + t_ItoC (seconds) */
    void append(double timestamp_inI, double load_img_ms, double prop_ms, double nn_ms, double update_ms, double total_ms) {
        if (!f_) return;
        std::fprintf(f_, "%.15f,%.5f,%.5f,%.5f,%.5f,%.5f\n", timestamp_inI, load_img_ms, prop_ms, nn_ms, update_ms, total_ms);
        std::fflush(f_);                       /* of_statistics.flush() */
    }
    void close() {
        if (f_) std::fclose(f_);
        f_ = nullptr;
    }

  private:
    /* mkdir -p of the file's directory; failures surface as a failed fopen */
    static void make_parent_dirs(const char* path) {
        std::string p(path);
        const size_t last = p.find_last_of('/');
        if (last == std::string::npos) return;
        for (size_t i = 1; i <= last; i++)
            if (p[i] == '/') {
                const std::string dir = p.substr(0, i);
                if (::mkdir(dir.c_str(), 0777) != 0 && errno != EEXIST) return;
            }
    }

    std::FILE* f_;
};

}  // namespace hnet_csv
#endif /* HNET_TIMING_CSV_H */
