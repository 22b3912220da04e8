// Launch wrappers implemented in rrtx_kernels.hip (the only translation unit with device code).
#ifndef RRTX_LAUNCH_H
#define RRTX_LAUNCH_H
#include <hip/hip_runtime_api.h>

#include "rrtx_device.h"

namespace rrtx {
template <typename F> hipError_t launch_render(const KernelParams<F> &P, bool filter, int lds_mode, int grid_blocks, hipStream_t stream);
template <typename F> hipError_t launch_primary_lists(const KernelParams<F> &P, uint16_t *plist, hipStream_t stream);
hipError_t launch_order_pixels(const uint16_t *plist, uint32_t n_pixels, uint32_t *scratch, uint32_t *order, hipStream_t stream); // KernelParams::pixel_order
template <typename F> hipError_t launch_sky_tasks(const KernelParams<F> &P, uint32_t first_position, uint32_t n_positions, int num_cus, hipStream_t stream); // the tasks of sky-only pixels
template <typename F> hipError_t launch_first_bounce(const KernelParams<F> &P, uint32_t n_positions, int num_cus, hipStream_t stream); // KernelParams::first
template <typename F> hipError_t launch_tail(const KernelParams<F> &P, bool filter, int grid_blocks, hipStream_t stream);
template <typename F> hipError_t launch_resume(const KernelParams<F> &P, bool filter, int grid_blocks, hipStream_t stream);
template <typename F> hipError_t launch_finalize(const F *partial, F *fb, const FinalizeShape &S, hipStream_t stream);
template <typename F> hipError_t launch_deinterleave(const F *gathered, F *frame, const GatherShape &S, hipStream_t stream);
template <typename F> hipError_t render_occupancy(const KernelParams<F> &P, bool filter, int lds_mode, int *blocks_per_cu);
} // namespace rrtx
#endif
