// Host side of csrc/s3_format.h (the weight packer of hnet_create runs it): the fp16 conversions against the compiler's own
// _Float16 conversion, and the error bounds of the two-plane formats.  Built host-only by tests/test_s3_format_host.py.
#include "cuahn_vio_amd/csrc/s3_format.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

int main() {
    using namespace hnet;
    std::mt19937_64 rng(1);
    long bad = 0, n = 0;
    auto chk = [&](float f) {
        const _Float16 h = (_Float16)f;
        uint16_t ref;
        memcpy(&ref, &h, 2);
        const uint16_t got = f32_to_f16_rn(f);
        if (got != ref && f == f) { if (bad < 10) printf("f32_to_f16_rn(%a) = %04x, compiler %04x\n", f, got, ref); bad++; }
        const float back = f16_to_f32(ref), rb = (float)h;
        if (memcmp(&back, &rb, 4) && rb == rb) { if (bad < 10) printf("f16_to_f32(%04x) = %a, compiler %a\n", ref, back, rb); bad++; }
        n++;
    };
    for (int i = 0; i < 4000000; i++) { uint32_t u = (uint32_t)rng(); float f; memcpy(&f, &u, 4); chk(f); }
    for (uint32_t h = 0; h < 65536; h++) {           // every fp16 value, its fp32 neighbours and the rounding ties between fp16 neighbours
        const float f = f16_to_f32((uint16_t)h);
        chk(f); chk(nextafterf(f, 1e30f)); chk(nextafterf(f, -1e30f));
        if ((h & 0x7FFF) < 0x7C00) chk(0.5f * (f + f16_to_f32((uint16_t)(h + 1))));
    }
    // activation split a = A0 + A1 / 4096: |error| <= max(2^-22 |a|, 2^-37)
    std::normal_distribution<float> nd(0.f, 1.f);
    std::mt19937 r2(3);
    double worst_a = 0, worst_w = 0, worst_g = 0;
    for (int i = 0; i < 2000000; i++) {
        const float a = nd(r2) * (i % 4 == 0 ? 1e-4f : i % 4 == 1 ? 1.f : i % 4 == 2 ? 300.f : 1e4f);     // |a| < 65504
        uint16_t x, y, z;
        split2h(a, x, y);
        worst_a = std::fmax(worst_a, std::fabs((double)join2h(x, y) - a) / std::fmax(std::ldexp(1.0, -22) * std::fabs(a), std::ldexp(1.0, -37)));
        // weight planes of the register-resident kernels: 4096 w = W0 + W1, W2 = W0 / 4096 (|w| < 16)
        const float w = nd(r2) * (i % 3 == 0 ? 1e-3f : i % 3 == 1 ? 0.05f : 3.f);
        if (std::fabs(w) < 15.9f) {
            wsplit2h(w, x, y, z);
            const double back = ((double)f16_to_f32(x) + (double)f16_to_f32(y)) / 4096.0;
            worst_w = std::fmax(worst_w, std::fabs(back - w) / std::fmax(std::ldexp(1.0, -22) * std::fabs(w), std::ldexp(1.0, -37)));
            worst_g = std::fmax(worst_g, std::fabs((double)f16_to_f32(z) * 4096.0 - (double)f16_to_f32(x)) / std::fmax(std::ldexp(1.0, -11) * std::fabs(f16_to_f32(x)), std::ldexp(1.0, -12)));
        }
    }
    // the RANGE of the activation split, exhaustively over every fp32 value of the bands concerned: both planes stay finite for
    // |a| < 32768 (proven bound: half an fp16 ulp there is <= 8, x 4096 = 32768 < 65520); in [32768, 65520) the first plane is still
    // finite but the residual of values within 2^-8 ulp of a rounding tie scales to >= 65520 and the SECOND plane becomes an infinity
    // (a non-finite result that the library detects, never a silently wrong one); from 65520 on the first plane overflows too.
    long inf_lo = 0, inf_mid = 0, fin_mid = 0, bad_err = 0;
    auto finite16 = [](uint16_t h) { return (h & 0x7C00u) != 0x7C00u; };
    for (uint32_t u = 0x46800000u; u < 0x47000000u; u++) {            // [16384, 32768): the top binade of the guaranteed range
        float a; memcpy(&a, &u, 4);
        uint16_t x, y; split2h(a, x, y);
        if (!finite16(x) || !finite16(y)) inf_lo++;
        else if (std::fabs((double)join2h(x, y) - a) > std::ldexp(1.0, -22) * a) bad_err++;
    }
    for (uint32_t u = 0x47000000u; u < 0x477FF000u; u++) {            // [32768, 65520)
        float a; memcpy(&a, &u, 4);
        uint16_t x, y; split2h(a, x, y);
        if (!finite16(x)) { inf_lo++; continue; }                      // must not happen below 65520
        if (!finite16(y)) inf_mid++; else { fin_mid++; if (std::fabs((double)join2h(x, y) - a) > std::ldexp(1.0, -22) * a) bad_err++; }
    }
    printf("activation range: [16384, 32768) non-finite planes %ld (must be 0); [32768, 65520): %ld values split finitely, %ld give an infinite second plane; "
           "error bound violations %ld\n", inf_lo, fin_mid, inf_mid, bad_err);
    if (inf_lo != 0 || bad_err != 0 || inf_mid == 0) return 1;       // inf_mid == 0 would mean the documented caveat is wrong
    printf("checked %ld conversions, %ld differ; worst error / bound: activation split %.3f, weight planes %.3f, W2 vs W0 %.3f\n", n, bad, worst_a, worst_w, worst_g);
    return (bad != 0 || worst_a > 1.0 || worst_w > 1.0 || worst_g > 1.0) ? 1 : 0;
}
