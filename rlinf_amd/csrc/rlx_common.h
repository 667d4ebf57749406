// Shared host/device helpers for librlx_hip.so (gfx950 / CDNA4 only -- no portability layer).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rlx.h"

#define RLX_WAVE 64

namespace rlx {

// ---- host-side error plumbing -------------------------------------------------------------
void set_error(const char* fmt, ...);

#define RLX_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::rlx::set_error(__VA_ARGS__);     \
            return RLX_EINVAL;                 \
        }                                      \
    } while (0)

#define RLX_HIP_CHECK(expr)                                                          \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            ::rlx::set_error("%s failed: %s", #expr, hipGetErrorString(_e));         \
            return RLX_EHIP;                                                         \
        }                                                                            \
    } while (0)

#define RLX_LAUNCH_CHECK()                                                           \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            ::rlx::set_error("kernel launch failed: %s", hipGetErrorString(_e));     \
            return RLX_EHIP;                                                         \
        }                                                                            \
    } while (0)

int num_cu();  // cached hipDeviceProp multiProcessorCount of the current device

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- device-side helpers --------------------------------------------------------------------
#if defined(__HIPCC__)

// Explicitly rounded f32 ops: the compiler may not contract these into FMAs, which keeps the
// sequential scans bit-identical to the reference's separate torch mul/add kernels.
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RLX_WAVE);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RLX_WAVE);
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RLX_WAVE);
    return v;
}

// Touch every 64-byte line of the first BYTES of the kernel-argument block at kernel start (scalar loads, all in flight together).
// A kernel whose argument structs span many lines otherwise pays one scalar-cache miss -- a trip to memory, microseconds when the
// chip is busy -- at every FIRST use of a new line, serially, wherever in the kernel that use happens to be.
template <int BYTES>
__device__ __forceinline__ void touch_kernargs() {
    typedef const __attribute__((address_space(4))) unsigned* kptr;
    kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned touch = 0;
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; ++i) touch |= kp[16 * i];
    asm volatile("" ::"s"(touch));
}

// Block-wide sum of K doubles per thread.  `scratch` must hold K * (blockDim.x/64) doubles.
// Result valid in thread 0 only.  Deterministic (fixed tree).
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    if (nw == 1) return;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) scratch[wid * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int w = 0; w < nw; ++w) s += scratch[w * K + k];
            v[k] = s;
        }
    }
}

#endif  // __HIPCC__

}  // namespace rlx
