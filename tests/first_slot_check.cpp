// CPU check of the first-bounce records' layout (rrt_amd/csrc/rrtx_device.h: first_slot, the code word's fields) against what rrtx_api.cpp allocates for them:
// for a sweep of (tasks, samples per task) every (task, sample, part) has a slot of its own inside ceil(tasks / 64) x 64 x samples x 2 slots of 4 F - no two records
// share one, none lies beyond the buffer -, the 64 tasks a wave holds lie side by side in each part (what makes the pre-pass's stores whole lines), and the code word
// keeps kind, draws (14 bits) and material (16 bits) apart.  No GPU, no HIP; built by tests/test_host.py with g++.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../rrt_amd/csrc/rrtx_device.h"
using namespace rrtx;
int main()
{
    long cases = 0;
    for (uint32_t tasks : {1u, 63u, 64u, 65u, 127u, 128u, 1000u, 4097u})
        for (uint32_t chunk : {1u, 2u, 7u, 8u, 16u, 33u}) {
            const size_t slots = (((size_t)tasks + 63) / 64) * 64 * (size_t)chunk * 2; // rrtx_api.cpp: bytes / (4 x sizeof(F))
            std::vector<unsigned char> seen(slots, 0);
            for (uint32_t t = 0; t < tasks; ++t)
                for (uint32_t k = 0; k < chunk; ++k)
                    for (uint32_t part = 0; part < 2; ++part) {
                        const size_t s = first_slot(t, k, chunk, part);
                        if (s >= slots || seen[s]) {
                            fprintf(stderr, "tasks %u chunk %u: slot %zu of (task %u, sample %u, part %u) %s\n", tasks, chunk, s, t, k, part, s >= slots ? "lies beyond the buffer" : "is taken");
                            return 1;
                        }
                        seen[s] = 1;
                        if ((t & 63u) != 0u && first_slot(t - 1, k, chunk, part) + 1 != s) {
                            fprintf(stderr, "tasks %u chunk %u: tasks %u and %u of one wave do not lie side by side\n", tasks, chunk, t - 1, t);
                            return 1;
                        }
                    }
            cases += 1;
        }
    // the code word: kind << 30 | draws << 16 | material
    for (uint32_t kind : {kFirstRay, kFirstDone, kFirstUnknown})
        for (uint32_t draws : {0u, 1u, 0x3FFFu})
            for (uint32_t mat : {0u, 1u, 0xFFFFu}) {
                const uint32_t code = (kind << 30) | (draws << 16) | mat;
                if ((code >> 30) != kind || ((code >> 16) & 0x3FFFu) != draws || (code & 0xFFFFu) != mat) {
                    fprintf(stderr, "code word: fields overlap (kind %u draws %u material %u)\n", kind, draws, mat);
                    return 1;
                }
            }
    if (kFirstRay == kFirstDone || kFirstDone == kFirstUnknown || kFirstRay == kFirstUnknown || kFirstUnknown > 3u) return 1;
    fprintf(stderr, "%ld layouts ok\n", cases);
    return 0;
}
