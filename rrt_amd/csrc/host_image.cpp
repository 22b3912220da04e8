// Host side of the seam: what main.cpp does with the framebuffer Rrt::render returns —
// the gamma-2 8-bit quantiser (color.h:8-23, rtweekend.h:93-98), the row flip
// (main.cpp:143,153), ASCII PPM to stdout (main.cpp:142-148) and a PNG file (main.cpp:164).
// The PNG encoder is our own (zlib deflate, filter 0; bands of rows deflated on several threads and joined
// into one zlib stream); decoded pixels are what matters.  SURVEY.md 8(f) N4: the encoder must not be what
// a batch of frames waits for - single-threaded deflate of a 1200x800 frame took longer than its render.
#include <zlib.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
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

// Threads one call may use: 8 at most, and no more than half of the CPUs this process is ALLOWED to use - a container's
// CPU quota (cgroup cpu.max), not the machine's 256 logical CPUs.  `rrt` keeps two writer tasks in flight beside the thread
// that drives the GPU; on a box with a quota of 16 CPUs, two times 16 deflate threads ran into the quota's throttling, which
// stalls every thread of the process, the one feeding the GPU included (a batch at the reference's animation settings:
// 9.1 ms per frame with 2 x 16 threads, 8.2 with 1 x 16, render 5.2).
unsigned host_threads()
{
    static const unsigned cached = [] {
        unsigned n = std::thread::hardware_concurrency();
        if (n == 0) n = 1;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota> <period>" in microseconds, or "max <period>"
            long long quota = 0, period = 0;
            if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
                const unsigned allowed = (unsigned)((quota + period - 1) / period);
                if (allowed < n) n = allowed;
            }
            std::fclose(f);
        }
        n /= 2;
        return n < 1 ? 1u : (n > 8 ? 8u : n);
    }();
    return cached;
}

// Work over n rows is cut into bands, one thread each: host_threads() at most, at least 256 KB of work per band
// (small images stay on the caller's thread).
int band_count(int n, size_t work_per_row)
{
    size_t bands = host_threads();
    const size_t by_work = (size_t)n * work_per_row / ((size_t)1 << 18);
    if (bands > by_work) bands = by_work;
    if (bands > (size_t)n) bands = (size_t)n;
    return bands < 2 ? 1 : (int)bands;
}
// body(first_row, last_row, band) for every band; band 0 on the calling thread
template <typename Body> void in_bands(int n, int bands, Body body)
{
    std::vector<std::thread> pool;
    int started = 1;
    try {
        for (; started < bands; ++started) pool.emplace_back(body, (int)((int64_t)n * started / bands), (int)((int64_t)n * (started + 1) / bands), started);
    }
    catch (const std::system_error &) { // no more threads to be had: the caller does the remaining bands itself
    }
    body(0, (int)((int64_t)n / bands), 0);
    for (int b = started; b < bands; ++b) body((int)((int64_t)n * b / bands), (int)((int64_t)n * (b + 1) / bands), b);
    for (std::thread &th : pool) th.join();
}

template <typename F> void quantise_image(const F *fb, int w, int h, int spp, uint8_t *rgb)
{
    const F scale = (F)1.0 / spp; // color.h:15
    in_bands(h, band_count(h, (size_t)w * 3 * 8), [=](int first, int last, int) {
        for (int out_row = first; out_row < last; ++out_row) {
            const F *src = fb + (size_t)(h - 1 - out_row) * w * 3; // main.cpp:153: top row = fb row h-1
            uint8_t *dst = rgb + (size_t)out_row * w * 3;
            for (int k = 0; k < w * 3; ++k) dst[k] = quantise_channel<F>(src[k], scale);
        }
    });
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
    // main.cpp:142 + color.h:31: "P3\nW H\n255\n" then one "r g b\n" line per pixel.  The text of a 4K frame is
    // ~ 90 MB: bands of rows are formatted on several threads from a table of the 256
    // decimal strings, then written in order.
    char dec[256][4];
    int dec_len[256];
    for (int v = 0; v < 256; ++v) dec_len[v] = std::snprintf(dec[v], sizeof dec[v], "%d", v);
    const int bands = band_count(h, (size_t)w * 3 * 16);
    std::vector<std::string> part((size_t)bands);
    in_bands(h, bands, [&](int first, int last, int b) {
        std::string &t = part[(size_t)b];
        t.resize((size_t)(last - first) * w * 12);
        char *o = &t[0];
        for (size_t p = (size_t)first * w; p < (size_t)last * w; ++p) {
            for (int k = 0; k < 3; ++k) {
                const uint8_t v = rgb[3 * p + k];
                std::memcpy(o, dec[v], 3); // (copies 3 bytes, advances by the true length)
                o += dec_len[v];
                *o++ = k == 2 ? '\n' : ' ';
            }
        }
        t.resize((size_t)(o - &t[0]));
    });
    std::string text = "P3\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
    size_t total = text.size();
    for (const std::string &t : part) total += t.size();
    text.reserve(total);
    for (const std::string &t : part) text += t;
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
    // Raw scanlines, each prefixed by filter type 0, deflated in bands of rows: every band is a raw deflate
    // stream of its own (no shared dictionary), all but the last ended with a full flush - byte aligned, not
    // final - so that their concatenation behind one zlib header is ONE valid stream; its Adler-32 is
    // combined from the bands'.
    const size_t stride = (size_t)w * 3 + 1;
    const int bands = band_count(h, stride * 8);
    std::vector<std::vector<uint8_t>> zb((size_t)bands);
    std::vector<uLong> adler((size_t)bands, 0);
    std::vector<size_t> raw_len((size_t)bands, 0);
    std::vector<int> ok_band((size_t)bands, 0);
    auto deflate_band = [&](int first, int last, int b) {
        std::vector<uint8_t> raw((size_t)(last - first) * stride);
        for (int y = first; y < last; ++y) {
            uint8_t *row = raw.data() + (size_t)(y - first) * stride;
            row[0] = 0;
            std::memcpy(row + 1, rgb + (size_t)y * w * 3, (size_t)w * 3);
        }
        // A path-traced frame is noise on top of the picture: string matching finds nothing in it and costs three
        // quarters of deflate's time, Huffman coding alone gets the same 0.85 (measured on 1280 x 720: 105 ms and
        // 0.848 with the default strategy, 31 ms and 0.845 with Z_RLE).  So: run-length + Huffman first, and only
        // a band that turns out to be compressible (a converged or flat image) is done again with matching, which
        // is quick on such data.
        std::vector<uint8_t> &out = zb[(size_t)b];
        bool fine = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, attempt == 0 ? Z_RLE : Z_DEFAULT_STRATEGY) != Z_OK) return;
            out.resize(deflateBound(&zs, (uLong)raw.size()) + 64);
            zs.next_in = raw.data(), zs.avail_in = (uInt)raw.size();
            zs.next_out = out.data(), zs.avail_out = (uInt)out.size();
            const int rc = deflate(&zs, b == bands - 1 ? Z_FINISH : Z_FULL_FLUSH);
            fine = (b == bands - 1 ? rc == Z_STREAM_END : rc == Z_OK) && zs.avail_in == 0;
            out.resize(out.size() - zs.avail_out);
            deflateEnd(&zs);
            if (!fine || out.size() * 10 > raw.size() * 6) break; // not below 0.6: matching would not help
        }
        adler[(size_t)b] = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
        raw_len[(size_t)b] = raw.size();
        ok_band[(size_t)b] = fine ? 1 : 0;
    };
    in_bands(h, bands, deflate_band);
    std::vector<uint8_t> z;
    z.push_back(0x78), z.push_back(0x9C); // zlib header: deflate, 32 K window, default level, no dictionary
    uLong sum = adler32(0L, Z_NULL, 0);
    for (int b = 0; b < bands; ++b) {
        if (!ok_band[(size_t)b]) return RRTX_E_IO;
        z.insert(z.end(), zb[(size_t)b].begin(), zb[(size_t)b].end());
        sum = b == 0 ? adler[0] : adler32_combine(sum, adler[(size_t)b], (z_off_t)raw_len[(size_t)b]);
    }
    put_be32(z, (uint32_t)sum);
    const size_t zlen = z.size();

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
