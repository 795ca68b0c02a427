// tests/cpp/ekf_check.cpp — drives include/hnet_ekf.h from a binary file of doubles (tests/test_ekf_cpu.py):
//   in : n_cases, then per case: p3 q4 v3 ba3 bg3 offset12 cov729 | mean8 cov64 propagated8 k_net_cov update_offset
//   out: per case: ok, p3 q4 v3 ba3 bg3 offset12 cov729, then the same after reset_4pt_offset
#include <cstdio>
#include <string>
#include <vector>
#include "hnet_ekf.h"

static void put(std::vector<double>& o, const hnet_ekf::State& s) {
    o.insert(o.end(), s.p, s.p + 3); o.insert(o.end(), s.q, s.q + 4); o.insert(o.end(), s.v, s.v + 3);
    o.insert(o.end(), s.ba, s.ba + 3); o.insert(o.end(), s.bg, s.bg + 3);
    o.insert(o.end(), &s.offset[0][0], &s.offset[0][0] + 12); o.insert(o.end(), s.cov, s.cov + 729);
}

// mode "prop": in: n, then per case p3 q4 v3 ba3 bg3 offset12 | c_R_i9 t3 dt w3 a3 ; out: p3 q4 v3 offset12
static int prop_mode(const char* fin, const char* fout) {
    FILE* f = std::fopen(fin, "rb");
    if (!f) return 2;
    double n;
    if (std::fread(&n, 8, 1, f) != 1) return 2;
    std::vector<double> out;
    for (int c = 0; c < (int)n; c++) {
        double in[28 + 9 + 3 + 1 + 3 + 3];
        if (std::fread(in, 8, sizeof in / 8, f) != sizeof in / 8) return 3;
        hnet_ekf::State s = {};
        const double* d = in;
        for (int i = 0; i < 3; i++) s.p[i] = *d++;
        for (int i = 0; i < 4; i++) s.q[i] = *d++;
        for (int i = 0; i < 3; i++) s.v[i] = *d++;
        for (int i = 0; i < 3; i++) s.ba[i] = *d++;
        for (int i = 0; i < 3; i++) s.bg[i] = *d++;
        for (int i = 0; i < 12; i++) (&s.offset[0][0])[i] = *d++;
        hnet_ekf::Extrinsics e;
        for (int i = 0; i < 9; i++) e.c_R_i[i] = *d++;
        for (int i = 0; i < 3; i++) e.i_t_i2c[i] = *d++;
        const double dt = *d++;
        const double* w = d; d += 3;
        const double* a = d; d += 3;
        hnet_ekf::propagate_mean(s, e, dt, w, a);
        out.insert(out.end(), s.p, s.p + 3); out.insert(out.end(), s.q, s.q + 4); out.insert(out.end(), s.v, s.v + 3);
        out.insert(out.end(), &s.offset[0][0], &s.offset[0][0] + 12);
    }
    std::fclose(f);
    FILE* g = std::fopen(fout, "wb");
    std::fwrite(out.data(), 8, out.size(), g);
    std::fclose(g);
    return 0;
}

// mode "jac": in: n, then per case p3 q4 v3 ba3 bg3 offset12 cov729 | c_R_i9 t3 dt w3 a3 qdiag15 ;
//             out: F729 Fw405, then the state after hnet_ekf::propagate (p3 q4 v3 ba3 bg3 offset12 cov729)
static int jac_mode(const char* fin, const char* fout) {
    FILE* f = std::fopen(fin, "rb");
    if (!f) return 2;
    double n;
    if (std::fread(&n, 8, 1, f) != 1) return 2;
    std::vector<double> out;
    for (int c = 0; c < (int)n; c++) {
        std::vector<double> in(757 + 9 + 3 + 1 + 3 + 3 + 15);
        if (std::fread(in.data(), 8, in.size(), f) != in.size()) return 3;
        hnet_ekf::State s;
        const double* d = in.data();
        for (int i = 0; i < 3; i++) s.p[i] = *d++;
        for (int i = 0; i < 4; i++) s.q[i] = *d++;
        for (int i = 0; i < 3; i++) s.v[i] = *d++;
        for (int i = 0; i < 3; i++) s.ba[i] = *d++;
        for (int i = 0; i < 3; i++) s.bg[i] = *d++;
        for (int i = 0; i < 12; i++) (&s.offset[0][0])[i] = *d++;
        for (int i = 0; i < 729; i++) s.cov[i] = *d++;
        hnet_ekf::Extrinsics e;
        for (int i = 0; i < 9; i++) e.c_R_i[i] = *d++;
        for (int i = 0; i < 3; i++) e.i_t_i2c[i] = *d++;
        const double dt = *d++;
        const double* w = d; d += 3;
        const double* a = d; d += 3;
        const double* q = d; d += 15;
        std::vector<double> F(729), Fw(405);
        hnet_ekf::propagate_jacobians(s, e, dt, w, F.data(), Fw.data());
        out.insert(out.end(), F.begin(), F.end());
        out.insert(out.end(), Fw.begin(), Fw.end());
        hnet_ekf::propagate(s, e, dt, w, a, q);
        put(out, s);
    }
    std::fclose(f);
    FILE* g = std::fopen(fout, "wb");
    std::fwrite(out.data(), 8, out.size(), g);
    std::fclose(g);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && std::string(argv[3]) == "prop") return prop_mode(argv[1], argv[2]);
    if (argc >= 4 && std::string(argv[3]) == "jac") return jac_mode(argv[1], argv[2]);
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    double n;
    if (std::fread(&n, 8, 1, f) != 1) return 2;
    std::vector<double> out;
    for (int c = 0; c < (int)n; c++) {
        double in[757 + 8 + 64 + 8 + 2];
        if (std::fread(in, 8, sizeof in / 8, f) != sizeof in / 8) return 3;
        hnet_ekf::State s;
        const double* d = in;
        for (int i = 0; i < 3; i++) s.p[i] = *d++;
        for (int i = 0; i < 4; i++) s.q[i] = *d++;
        for (int i = 0; i < 3; i++) s.v[i] = *d++;
        for (int i = 0; i < 3; i++) s.ba[i] = *d++;
        for (int i = 0; i < 3; i++) s.bg[i] = *d++;
        for (int i = 0; i < 12; i++) (&s.offset[0][0])[i] = *d++;
        for (int i = 0; i < 729; i++) s.cov[i] = *d++;
        const double* mean = d; d += 8;
        const double* cov = d; d += 64;
        const double* prop = d; d += 8;
        const double k = *d++;
        const bool upd = *d++ != 0.0;
        const bool ok = hnet_ekf::update(s, mean, cov, prop, k, upd);
        out.push_back(ok ? 1.0 : 0.0);
        put(out, s);
        hnet_ekf::reset_4pt_offset(s);
        put(out, s);
    }
    std::fclose(f);
    FILE* g = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), 8, out.size(), g);
    std::fclose(g);
    return 0;
}
