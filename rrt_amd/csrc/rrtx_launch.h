// Launch wrappers implemented in rrtx_kernels.hip (the only translation unit with device code).
#ifndef RRTX_LAUNCH_H
#define RRTX_LAUNCH_H
#include <hip/hip_runtime_api.h>

#include "rrtx_device.h"

namespace rrtx {
template <typename F> hipError_t launch_render(const KernelParams<F> &P, bool filter, int lds_mode, int grid_blocks, hipStream_t stream);
template <typename F> hipError_t launch_primary_lists(const KernelParams<F> &P, uint16_t *plist, hipStream_t stream);
template <typename F> hipError_t launch_tail(const KernelParams<F> &P, bool filter, int grid_blocks, hipStream_t stream);
template <typename F> hipError_t launch_resume(const KernelParams<F> &P, bool filter, int grid_blocks, hipStream_t stream);
template <typename F> hipError_t launch_finalize(const F *partial, F *fb, const FinalizeShape &S, hipStream_t stream);
template <typename F> hipError_t launch_deinterleave(const F *gathered, F *frame, const GatherShape &S, hipStream_t stream);
template <typename F> hipError_t render_occupancy(const KernelParams<F> &P, bool filter, int lds_mode, int *blocks_per_cu);
} // namespace rrtx
#endif
