// Host side of the seam: what main.cpp does with the framebuffer Rrt::render returns —
// the gamma-2 8-bit quantiser (color.h:8-23, rtweekend.h:93-98), the row flip
// (main.cpp:143,153), ASCII PPM to stdout (main.cpp:142-148) and a PNG file (main.cpp:164).
// The PNG encoder is our own (zlib deflate, filter 0); decoded pixels are what matters.
#include <zlib.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rrtx.h"

namespace {

inline float root(float x) { return sqrtf(x); } // SQRT = sqrtf for FP_T = float (rtweekend.h:23)
inline double root(double x) { return sqrt(x); }

// color.h:16-22: static_cast<int>(256 * clamp(v, 0, 0.999)), clamp() returning *double*
// (rtweekend.h:93).  A NaN passes clamp() unchanged; rrtc's x86-64 conversion then yields
// INT_MIN, whose low byte is 0.  Spelled out so that no undefined conversion is executed.
template <typename F> inline uint8_t quantise_channel(F sum, F scale)
{
    const F v = root(scale * sum);
    const F lo = (F)0.0, hi = (F)0.999;
    double clamped;
    if (v < lo)
        clamped = lo;
    else if (v > hi)
        clamped = hi;
    else
        clamped = v;
    const double scaled = 256 * clamped;
    int as_int;
    if (scaled > -2147483649.0 && scaled < 2147483648.0)
        as_int = (int)scaled;
    else
        as_int = INT_MIN; // NaN or out of range
    return (uint8_t)as_int;
}

template <typename F> void quantise_image(const F *fb, int w, int h, int spp, uint8_t *rgb)
{
    const F scale = (F)1.0 / spp; // color.h:15
    for (int out_row = 0; out_row < h; ++out_row) {
        const F *src = fb + (size_t)(h - 1 - out_row) * w * 3; // main.cpp:153: top row = fb row h-1
        uint8_t *dst = rgb + (size_t)out_row * w * 3;
        for (int k = 0; k < w * 3; ++k) dst[k] = quantise_channel<F>(src[k], scale);
    }
}

void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}

void put_chunk(std::vector<uint8_t> &file, const char tag[4], const uint8_t *data, size_t n)
{
    put_be32(file, (uint32_t)n);
    const size_t start = file.size();
    file.insert(file.end(), tag, tag + 4);
    if (n) file.insert(file.end(), data, data + n);
    put_be32(file, (uint32_t)crc32(0L, file.data() + start, (uInt)(n + 4)));
}

} // namespace

extern "C" {

int rrtx_quantise(const void *fb, int fp64, int w, int h, int spp, uint8_t *rgb)
{
    if (!fb || !rgb || w < 1 || h < 1 || spp < 1) return RRTX_E_INVALID;
    if (fp64)
        quantise_image<double>((const double *)fb, w, h, spp, rgb);
    else
        quantise_image<float>((const float *)fb, w, h, spp, rgb);
    return RRTX_OK;
}

int rrtx_write_ppm(const char *path, const uint8_t *rgb, int w, int h)
{
    if (!rgb || w < 1 || h < 1) return RRTX_E_INVALID;
    FILE *f = (!path || std::strcmp(path, "-") == 0) ? stdout : std::fopen(path, "w");
    if (!f) return RRTX_E_IO;
    // main.cpp:142 + color.h:31: "P3\nW H\n255\n" then one "r g b\n" line per pixel
    std::string text;
    text.reserve((size_t)w * h * 12 + 32);
    text += "P3\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
    char line[16];
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        int n = std::snprintf(line, sizeof line, "%u %u %u\n", rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2]);
        text.append(line, (size_t)n);
    }
    const bool ok = std::fwrite(text.data(), 1, text.size(), f) == text.size();
    if (f != stdout)
        std::fclose(f);
    else
        std::fflush(f);
    return ok ? RRTX_OK : RRTX_E_IO;
}

int rrtx_write_png(const char *path, const uint8_t *rgb, int w, int h)
{
    if (!path || !rgb || w < 1 || h < 1) return RRTX_E_INVALID;
    // raw scanlines, each prefixed by filter type 0
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * 3 + 1));
    for (int y = 0; y < h; ++y) {
        uint8_t *row = raw.data() + (size_t)y * ((size_t)w * 3 + 1);
        row[0] = 0;
        std::memcpy(row + 1, rgb + (size_t)y * w * 3, (size_t)w * 3);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return RRTX_E_IO;

    std::vector<uint8_t> file;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    file.insert(file.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w);
    put_be32(ihdr, (uint32_t)h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0}; // 8-bit, truecolour RGB, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(file, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(file, "IDAT", z.data(), zlen);
    put_chunk(file, "IEND", nullptr, 0);

    FILE *f = std::fopen(path, "wb");
    if (!f) return RRTX_E_IO;
    const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
    std::fclose(f);
    return ok ? RRTX_OK : RRTX_E_IO;
}

} // extern "C"
