// Wave64 scans over the 64 lanes with DPP row shifts and row broadcasts (gfx9 family: row_shr within the 16-lane rows,
// row_bcast:15 / row_bcast:31 across them) - six VALU instructions each, no LDS.  Used by the dense (ray, entry) pairing of the
// accelerated closest hit (rrtx_kernels.hip, dense_candidates); tools/ubench_scan.hip checks them against a plain loop.
#ifndef RRTX_WAVE_H
#define RRTX_WAVE_H
#include <stdint.h>

namespace rrtx {

// inclusive prefix sum: lane l gets v[0] + ... + v[l]
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return v;
}
// inclusive prefix maximum (unsigned; identity 0)
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}

} // namespace rrtx
#endif
