// tests/cpp/b41_tap_check.cpp — compile-time properties of the block_4_1 tap table of the fused block-4 kernel (csrc/kernels.h b41_tap, shared by the kernel's address
// setup and hnet_create's fragment packing): every one of the 25 taps exactly once; the two lane groups of a shared ds_read_b128 hold taps of ONE kernel column (their
// pixels are then whole image rows = multiples of 256 bytes apart: conflict-free) or one of them holds none - except the last pair of step 6.
#include "cuahn_vio_amd/csrc/kernels.h"
using namespace hnet;

constexpr bool table_ok() {
    int seen[25] = {};
    for (int st = 0; st < 7; st++)
        for (int g = 0; g < 4; g++) {
            const int t = b41_tap(st, g);
            if (t < -1 || t >= 25) return false;
            if (t >= 0) seen[t]++;
            if ((g & 1) && t >= 0 && b41_tap(st, g - 1) >= 0 && (t % 5) != (b41_tap(st, g - 1) % 5) && !(st == 6 && g == 3)) return false;
        }
    for (int t = 0; t < 25; t++)
        if (seen[t] != 1) return false;
    return true;
}
static_assert(table_ok(), "b41_tap: every tap once, shared reads within one kernel column");
static_assert(B42_HP >= 112 + B42_PADY && B42_WP >= 160 + B42_PADX + 1, "the bordered block_4_1 map holds the image and the patch of the last tile (pixel 35 of its rows)");
static_assert((B42_WP * 32) % 128 == 0, "rows of the bordered map start on 128-byte lines");
int main() { return 0; }
