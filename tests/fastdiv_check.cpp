// Host check of rrtx::make_fastdiv (rrt_amd/csrc/rrtx_device.h): n / d == umulhi(n, m) >> shift for n < 2^31.
#include <cstdio>
#include <cstdlib>
#include "../rrt_amd/csrc/rrtx_device.h"
static uint32_t fdiv(uint32_t n, const rrtx::FastDiv &f) { return f.is_one ? n : (uint32_t)(((uint64_t)n * f.m) >> 32) >> f.shift; }
int main()
{
    uint64_t state = 88172645463325252ull, bad = 0, checked = 0;
    auto rnd = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    auto check = [&](uint32_t d) {
        const rrtx::FastDiv f = rrtx::make_fastdiv(d);
        const uint32_t edge[] = {0u, 1u, d - 1u, d, d + 1u, 2u * d - 1u, 2u * d, 0x7FFFFFFFu, 0x7FFFFFFEu, 0x7FFFFFFFu / d * d, 0x7FFFFFFFu / d * d - 1u};
        for (uint32_t n : edge)
            if (n < 0x80000000u) { checked++; if (fdiv(n, f) != n / d) bad++; }
        for (int i = 0; i < 200; ++i) { uint32_t n = (uint32_t)rnd() & 0x7FFFFFFFu; checked++; if (fdiv(n, f) != n / d) bad++; }
    };
    for (uint32_t d = 1; d <= 70000; ++d) check(d);
    for (int i = 0; i < 200000; ++i) { uint32_t d = (uint32_t)rnd() & 0x7FFFFFFFu; if (d) check(d); }
    for (int L = 1; L < 31; ++L) for (int off = -2; off <= 2; ++off) if ((1u << L) + off > 0) check((1u << L) + off);
    std::printf("checked %llu bad %llu\n", (unsigned long long)checked, (unsigned long long)bad);
    return bad != 0;
}
