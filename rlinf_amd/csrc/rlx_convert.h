// rlx_convert.h -- dtype conversion and wide-access helpers shared by the weight-sync kernels (weight_patch.hip,
// bucket_copy.hip).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rlx {

// ---- element conversion: SRC = dtype read, DST = dtype written -------------------------------------------------------
template <typename SRC, typename DST>
struct Conv {
    static __device__ __forceinline__ DST cvt(SRC v) { return (DST)v; }
};
template <typename T>
struct Conv<T, T> {
    static __device__ __forceinline__ T cvt(T v) { return v; }
};
// narrowing follows c10's converters bit for bit: round to nearest even, and every NaN becomes the canonical quiet NaN
// (bf16 0x7FC0; f16 sign | 0x7E00), so that the value bytes on the wire equal the reference's
template <>
struct Conv<float, __bf16> {
    static __device__ __forceinline__ __bf16 cvt(float v) {
        if (v != v) {
            const uint16_t q = 0x7FC0;
            return *reinterpret_cast<const __bf16*>(&q);
        }
        return (__bf16)v;
    }
};
template <>
struct Conv<float, _Float16> {
    static __device__ __forceinline__ _Float16 cvt(float v) {
        if (v != v) {
            const uint16_t q = (uint16_t)(((__float_as_uint(v) >> 16) & 0x8000u) | 0x7E00u);
            return *reinterpret_cast<const _Float16*>(&q);
        }
        return (_Float16)v;
    }
};
// 16-bit float to the other 16-bit float: through f32, as torch's copy kernels do
template <>
struct Conv<__bf16, _Float16> {
    static __device__ __forceinline__ _Float16 cvt(__bf16 v) { return Conv<float, _Float16>::cvt((float)v); }
};
template <>
struct Conv<_Float16, __bf16> {
    static __device__ __forceinline__ __bf16 cvt(_Float16 v) { return Conv<float, __bf16>::cvt((float)v); }
};

template <typename T, int N>
struct alignas((sizeof(T) * N) > 16 ? 16 : (sizeof(T) * N)) Pack {
    T v[N];
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// BYTES (8, 16, 32 or 64) of one lane's 8 elements, fetched with streaming (non-temporal) 16-byte loads
template <int BYTES>
struct RawPack {
    uint32_t w[BYTES / 4];
    __device__ __forceinline__ void load(const void* p) {
        if constexpr (BYTES == 8) {
            const u32x2 q = __builtin_nontemporal_load(static_cast<const u32x2*>(p));
            w[0] = q.x, w[1] = q.y;
        } else {
#pragma unroll
            for (int j = 0; j < BYTES / 16; ++j) {
                const u32x4 q = __builtin_nontemporal_load(static_cast<const u32x4*>(p) + j);
                w[4 * j] = q.x, w[4 * j + 1] = q.y, w[4 * j + 2] = q.z, w[4 * j + 3] = q.w;
            }
        }
    }
};

}  // namespace rlx
