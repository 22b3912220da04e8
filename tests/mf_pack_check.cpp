// Host check of the matrix-core filter's sphere table (rrtx_pack.h: pack_mf_table), driven by tests/test_filter_mfma.py:
// reads n and n x {cx cy cz r2} (float32 bit patterns, hex) from stdin, prints per sphere its 32 f16 terms (hex) in TERM order
// - undoing the [block of 32][half of the terms][lane][8] layout the same way the kernel's lanes read it - and whether it is listed apart.
#include <cstdio>
#include <cstdlib>

#include "../rrt_amd/csrc/rrtx_pack.h"

int main()
{
    int n = 0;
    if (scanf("%d", &n) != 1 || n < 1) return 2;
    const int n_pad = (n + rrtx::kSpherePad - 1) / rrtx::kSpherePad * rrtx::kSpherePad;
    std::vector<rrtx::SphereHot<float>> hot((size_t)n_pad);
    for (int i = 0; i < n; ++i) {
        unsigned v[4];
        if (scanf("%x %x %x %x", &v[0], &v[1], &v[2], &v[3]) != 4) return 2;
        memcpy(&hot[i].cx, &v[0], 4), memcpy(&hot[i].cy, &v[1], 4), memcpy(&hot[i].cz, &v[2], 4), memcpy(&hot[i].r2, &v[3], 4);
    }
    rrtx::MfTable mf;
    rrtx::pack_mf_table<float>(hot, n, n_pad, mf);
    const int n_tab = (int)(mf.halves.size() / 32); // whole blocks of 32 spheres
    printf("%d %d %d\n", n_tab, (int)mf.big.size(), mf.ok ? 1 : 0);
    std::vector<char> apart((size_t)n_tab, 0);
    for (uint32_t b : mf.big) apart[b] = 1;
    for (int i = 0; i < n_tab; ++i) {
        printf("%d", (int)apart[i]);
        for (int kh = 0; kh < 2; ++kh)             // the kernel: lane l reads 16 bytes at (((i / 32) * 2 + kh) * 64 + l) * 16, l = 32 * lane_hi + i % 32,
            for (int lane_hi = 0; lane_hi < 2; ++lane_hi) // and finds terms 16 kh + 8 lane_hi ... + 7 of sphere i there
                for (int j = 0; j < 8; ++j) printf(" %04x", mf.halves[((((size_t)(i / 32) * 2 + (size_t)kh) * 64 + (size_t)(32 * lane_hi + i % 32)) * 8) + (size_t)j]);
        printf("\n");
    }
    // the f16 conversions themselves, on a few thousand values across the range (both directions)
    unsigned bad = 0;
    for (uint32_t h = 0; h < 65536; ++h) {
        const float f = rrtx::f16_bits_to_f32((uint16_t)h);
        if (f != f) continue;
        if (rrtx::f32_to_f16_bits(f) != h) bad += 1;
    }
    printf("roundtrip_bad %u\n", bad);
    return 0;
}
