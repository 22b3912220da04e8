// CPU check of the multi-device row-tile plan (rrt_amd/csrc/rrtx_device.h: shard_row_count, shard_local_to_frame_row,
// gather_source_row - the same source the kernels and rrtx_group.cpp compile): for every (H, T, N) of a sweep
//   * the shards' row sets are disjoint and cover the frame, in ascending order per shard, and their sizes are
//     shard_row_count's;
//   * a gathered buffer built the way rrtx_group.cpp builds it (shard r's compact block from row_off[r] on, its local
//     row lr holding frame row shard_local_to_frame_row(lr)) comes back as the identity through gather_source_row -
//     what deinterleave_kernel computes per value;
// and prints one line per case, "H T N : rows of shard 0 | rows of shard 1 | ...", which tests/test_dist_cpu.py holds
// against rrt_amd/dist.py's shard_rows (the rule the one-process-per-GPU path and bench.py use).
// No GPU, no HIP; built by the test with g++.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../rrt_amd/csrc/rrtx_device.h"
using namespace rrtx;
int main(int argc, char **argv)
{
    const bool print = argc > 1;
    long cases = 0;
    for (uint32_t N = 1; N <= 9; ++N)
        for (uint32_t T : {1u, 2u, 3u, 4u, 5u, 8u, 16u})
            for (uint32_t H : {2u, 3u, 4u, 7u, 8u, 9u, 15u, 16u, 17u, 31u, 33u, 64u, 100u, 266u, 800u, 2160u}) {
                GatherShape S = {};
                S.row_values = 3, S.height = H, S.tile_rows = T, S.n_shards = N;
                std::vector<uint32_t> gathered; // frame row held by each row of the gathered buffer
                std::vector<int> seen(H, 0);
                uint32_t off = 0;
                if (print) printf("%u %u %u :", H, T, N);
                for (uint32_t r = 0; r < N; ++r) {
                    S.row_off[r] = off;
                    const uint32_t n = shard_row_count(H, T, N, r);
                    // the definition (rrtx_api.cpp rows_of_shard / rrtx.h): row j belongs to shard (j / T) mod N
                    uint32_t by_definition = 0;
                    for (uint32_t j = 0; j < H; ++j) by_definition += (j / T) % N == r;
                    if (n != by_definition) return fprintf(stderr, "H %u T %u N %u shard %u: count %u, definition %u\n", H, T, N, r, n, by_definition), 1;
                    uint32_t prev = 0;
                    for (uint32_t lr = 0; lr < n; ++lr) {
                        const uint32_t j = shard_local_to_frame_row(lr, T, N, r);
                        if (j >= H || (j / T) % N != r || (lr && j <= prev)) return fprintf(stderr, "H %u T %u N %u shard %u: local row %u -> %u\n", H, T, N, r, lr, j), 1;
                        seen[j] += 1, prev = j;
                        gathered.push_back(j);
                        if (print) printf(" %u", j);
                    }
                    if (print) printf(r + 1 < N ? " |" : "\n");
                    off += n;
                }
                if (off != H) return fprintf(stderr, "H %u T %u N %u: %u rows in all\n", H, T, N, off), 1;
                for (uint32_t j = 0; j < H; ++j) {
                    if (seen[j] != 1) return fprintf(stderr, "H %u T %u N %u: row %u rendered %d times\n", H, T, N, j, seen[j]), 1;
                    const uint32_t g = gather_source_row(S, j);
                    if (g >= H || gathered[g] != j) return fprintf(stderr, "H %u T %u N %u: frame row %u read from gathered row %u\n", H, T, N, j, g), 1;
                }
                cases += 1;
            }
    fprintf(stderr, "gather plan: %ld cases ok\n", cases);
    return 0;
}
