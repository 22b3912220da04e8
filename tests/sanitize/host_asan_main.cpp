// Driver for the sanitizer run of the host code (tests/test_host.py): parser on good and bad scene files,
// quantiser and image writers, built together with host_scene.cpp / host_image.cpp under ASan + UBSan.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/rrtx.h"
int main(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i)
        for (int fp64 = 0; fp64 < 2; ++fp64) {
            rrtx_scene *s = nullptr;
            int rc = rrtx_scene_load(argv[i], 120, 80, fp64, &s);
            std::printf("%s fp64=%d rc=%d exit=%d\n", argv[i], fp64, rc, rrtx_scene_exit_code());
            if (!rc) {
                rrtx_scene_desc d;
                rrtx_scene_describe(s, &d);
                int32_t c[6];
                rrtx_scene_counts(s, c);
                rrtx_scene_free(s);
            }
        }
    std::vector<float> fb(64 * 48 * 3);
    for (size_t i = 0; i < fb.size(); ++i) fb[i] = (float)(i % 97) / 10.0f;
    fb[5] = -1.0f, fb[7] = 1e30f, fb[9] = 0.0f / 1.0f;
    std::vector<uint8_t> rgb(fb.size());
    rrtx_quantise(fb.data(), 0, 64, 48, 4, rgb.data());
    std::printf("png %d ppm %d\n", rrtx_write_png("asan.png", rgb.data(), 64, 48), rrtx_write_ppm("asan.ppm", rgb.data(), 64, 48));
    return 0;
}
