// Property check of rrtx_grid.h's triangle_touches_box (tests/test_host.py): it may say "touches" too often, never too
// rarely.  Random triangles and boxes, from needles to slabs; whenever a point sampled on the triangle lies in the box
// the function must answer true, and boxes far outside the triangle's bounding box must be rejected (so that it is not
// a constant).  argv[1]: cases.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../rrt_amd/csrc/rrtx_device.h"
#include "../rrt_amd/csrc/rrtx_grid.h"
int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 200000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(-1.0, 1.0), U01(0.0, 1.0);
    long touched = 0, rejected = 0, wrong = 0;
    for (int c = 0; c < cases; ++c) {
        double v[3][3], lo[3], hi[3];
        const double scale = std::pow(10.0, 2.0 * U(rng)), thin = c % 5 == 0 ? 1e-6 : 1.0;
        for (int q = 0; q < 3; ++q)
            for (int k = 0; k < 3; ++k) v[q][k] = scale * U(rng) * (k == 2 ? thin : 1.0);
        if (c % 7 == 0)
            for (int k = 0; k < 3; ++k) v[2][k] = v[0][k] + 0.5 * (v[1][k] - v[0][k]); // degenerate: collinear
        {
            // half of the boxes lie near a point of the triangle (touching, grazing, just missing), the rest anywhere
            double a = U01(rng), b = U01(rng);
            if (a + b > 1) a = 1 - a, b = 1 - b;
            const double size = scale * std::pow(10.0, 2.0 * U(rng) - 1.5);
            for (int k = 0; k < 3; ++k) {
                const double pk = v[0][k] + a * (v[1][k] - v[0][k]) + b * (v[2][k] - v[0][k]);
                const double ck = c % 2 == 0 ? pk + size * 1.2 * U(rng) : scale * U(rng);
                lo[k] = ck - 0.5 * size * U01(rng), hi[k] = ck + 0.5 * size * U01(rng);
            }
        }
        const bool says = rrtx::triangle_touches_box(v, lo, hi, 0.0);
        bool inside = false;
        for (int s = 0; s < 64 && !inside; ++s) {
            double a = U01(rng), b = U01(rng);
            if (a + b > 1) a = 1 - a, b = 1 - b;
            bool in = true;
            for (int k = 0; k < 3; ++k) {
                const double pk = v[0][k] + a * (v[1][k] - v[0][k]) + b * (v[2][k] - v[0][k]);
                in = in && pk >= lo[k] && pk <= hi[k];
            }
            inside = in;
        }
        if (inside && !says) wrong += 1;
        touched += says, rejected += !says;
    }
    std::printf("cases %d touched %ld rejected %ld wrong %ld\n", cases, touched, rejected, wrong);
    return wrong ? 1 : (rejected == 0 || touched == 0 ? 2 : 0);
}
