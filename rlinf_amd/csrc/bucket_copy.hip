// bucket_copy.hip -- table-driven multi-tensor cast + copy for the bucket weight syncer, gfx950.
//
// Replaces the per-tensor `tensor.to(device=bucket_device, dtype=transport_dtype)` chain of iter_named_tensor_buckets
// (rlinf/hybrid_engines/weight_syncer/bucket_syncer.py:110-121: one cast kernel + one allocation per parameter) and, on
// the receiver, the per-parameter `param.copy_(input_param)` of load_state_dict (:296-323): ONE launch moves every
// tensor of a bucket between its own storage and one flat transport buffer, converting on the way.
//
//   segment = (src, dst, n elements, src dtype, dst dtype); a workgroup owns one 4096-element chunk of one segment and
//   finds it by a binary search over the segments' first-chunk numbers (the table sits in L2 after the first wave).
//   Interior chunks with 16-byte aligned ends: every lane issues its two 8-element loads before the first use, converts
//   in registers, writes 16/32-byte packs.  HBM-bound: 4 B read + 2 B written per element for f32 -> bf16.
//
// Conversions are c10's (round to nearest even, canonical NaN), so the transport bytes equal what torch's .to() makes.

#include "rlx_common.h"
#include "rlx_convert.h"

#ifndef RLX_COPY_NT_STORE
#define RLX_COPY_NT_STORE 0  // dev switch: streaming stores into the transport buffer
#endif

namespace rlx {
namespace {

constexpr int CT = 256;
constexpr int LANE = 8;                       // elements per lane per load
constexpr int ITERS = RLX_COPY_CHUNK / (CT * LANE);
static_assert(ITERS * CT * LANE == RLX_COPY_CHUNK, "chunk = whole iterations of the block");

// The table hands out generic pointers; every one of them is device (global) memory, and saying so gets global_load /
// global_store instead of flat_* (which also tick the LDS counter and cannot be streamed).
#define RLX_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));

template <int BYTES>
__device__ __forceinline__ void load_lane(uint32_t (&w)[BYTES / 4], const void* p) {
    if constexpr (BYTES == 8) {
        const u32x2 q = __builtin_nontemporal_load((const RLX_GLOBAL u32x2*)p);
        w[0] = q.x, w[1] = q.y;
    } else {
#pragma unroll
        for (int j = 0; j < BYTES / 16; ++j) {
            const u32x4 q = __builtin_nontemporal_load((const RLX_GLOBAL u32x4*)p + j);
            w[4 * j] = q.x, w[4 * j + 1] = q.y, w[4 * j + 2] = q.z, w[4 * j + 3] = q.w;
        }
    }
}
template <int BYTES>
__device__ __forceinline__ void store_lane(void* p, const uint32_t (&w)[BYTES / 4]) {
    if constexpr (BYTES == 8) {
        u32x2 q;
        q.x = w[0], q.y = w[1];
#if RLX_COPY_NT_STORE
        __builtin_nontemporal_store(q, (RLX_GLOBAL u32x2*)p);
#else
        *(RLX_GLOBAL u32x2*)p = q;
#endif
    } else {
#pragma unroll
        for (int j = 0; j < BYTES / 16; ++j) {
            u32x4 q;
            q.x = w[4 * j], q.y = w[4 * j + 1], q.z = w[4 * j + 2], q.w = w[4 * j + 3];
#if RLX_COPY_NT_STORE
            __builtin_nontemporal_store(q, (RLX_GLOBAL u32x4*)p + j);
#else
            *((RLX_GLOBAL u32x4*)p + j) = q;
#endif
        }
    }
}

template <typename SRC, typename DST>
__device__ __forceinline__ void copy_chunk(const void* src_, void* dst_, long long e0, long long e1) {
    const RLX_GLOBAL SRC* __restrict__ src = (const RLX_GLOBAL SRC*)src_;
    RLX_GLOBAL DST* __restrict__ dst = (RLX_GLOBAL DST*)dst_;
    const bool aligned = ((reinterpret_cast<uintptr_t>(src_) | reinterpret_cast<uintptr_t>(dst_)) & 15) == 0;
    if (aligned && e1 - e0 == RLX_COPY_CHUNK) {  // block-uniform: no per-lane guards, all loads in flight at once
        uint32_t raw[ITERS][sizeof(SRC) * LANE / 4];
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
            load_lane<sizeof(SRC) * LANE>(raw[i], static_cast<const SRC*>(src_) + e0 + (long long)(i * CT + threadIdx.x) * LANE);
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const SRC* a = reinterpret_cast<const SRC*>(raw[i]);
            union {
                DST v[LANE];
                uint32_t w[sizeof(DST) * LANE / 4];
            } out;
#pragma unroll
            for (int j = 0; j < LANE; ++j) out.v[j] = Conv<SRC, DST>::cvt(a[j]);
            store_lane<sizeof(DST) * LANE>(static_cast<DST*>(dst_) + e0 + (long long)(i * CT + threadIdx.x) * LANE, out.w);
        }
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += CT) dst[e] = Conv<SRC, DST>::cvt(src[e]);
    }
}

// dtype pair -> one code, block-uniform switch in the kernel; same-dtype copies are raw by element size
enum Path { P_RAW8, P_RAW16, P_RAW32, P_RAW64, P_F32_BF16, P_F32_F16, P_BF16_F32, P_F16_F32, P_BF16_F16, P_F16_BF16, P_BAD };

__host__ __device__ inline int dtype_bytes(int d) {
    switch (d) {
        case RLX_DTYPE_F32: case RLX_DTYPE_RAW32: return 4;
        case RLX_DTYPE_BF16: case RLX_DTYPE_F16: case RLX_DTYPE_RAW16: return 2;
        case RLX_DTYPE_RAW8: return 1;
        case RLX_DTYPE_RAW64: return 8;
        default: return 0;
    }
}

__host__ __device__ inline int path_of(int s, int d) {
    const int sb = dtype_bytes(s), db = dtype_bytes(d);
    if (sb == 0 || db == 0) return P_BAD;
    if (s == d) return sb == 1 ? P_RAW8 : sb == 2 ? P_RAW16 : sb == 4 ? P_RAW32 : P_RAW64;
    if (s == RLX_DTYPE_F32 && d == RLX_DTYPE_BF16) return P_F32_BF16;
    if (s == RLX_DTYPE_F32 && d == RLX_DTYPE_F16) return P_F32_F16;
    if (s == RLX_DTYPE_BF16 && d == RLX_DTYPE_F32) return P_BF16_F32;
    if (s == RLX_DTYPE_F16 && d == RLX_DTYPE_F32) return P_F16_F32;
    if (s == RLX_DTYPE_BF16 && d == RLX_DTYPE_F16) return P_BF16_F16;
    if (s == RLX_DTYPE_F16 && d == RLX_DTYPE_BF16) return P_F16_BF16;
    return P_BAD;  // raw (integer / bool / f64) data never changes dtype on this path
}

__global__ __launch_bounds__(CT) void copy_segments_kernel(const rlx_copy_segment* __restrict__ table, int n_segments) {
    const long long chunk = blockIdx.x;
    int lo = 0, hi = n_segments;  // last segment whose first_chunk <= chunk (empty segments share their successor's number)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (table[mid].first_chunk <= chunk) lo = mid; else hi = mid;
    }
    const rlx_copy_segment seg = table[lo];
    const long long e0 = (chunk - seg.first_chunk) * RLX_COPY_CHUNK;
    const long long e1 = e0 + RLX_COPY_CHUNK < seg.n ? e0 + RLX_COPY_CHUNK : seg.n;
    switch (path_of(seg.src_dtype, seg.dst_dtype)) {
        case P_RAW8: copy_chunk<uint8_t, uint8_t>(seg.src, seg.dst, e0, e1); break;
        case P_RAW16: copy_chunk<uint16_t, uint16_t>(seg.src, seg.dst, e0, e1); break;
        case P_RAW32: copy_chunk<uint32_t, uint32_t>(seg.src, seg.dst, e0, e1); break;
        case P_RAW64: copy_chunk<unsigned long long, unsigned long long>(seg.src, seg.dst, e0, e1); break;
        case P_F32_BF16: copy_chunk<float, __bf16>(seg.src, seg.dst, e0, e1); break;
        case P_F32_F16: copy_chunk<float, _Float16>(seg.src, seg.dst, e0, e1); break;
        case P_BF16_F32: copy_chunk<__bf16, float>(seg.src, seg.dst, e0, e1); break;
        case P_F16_F32: copy_chunk<_Float16, float>(seg.src, seg.dst, e0, e1); break;
        case P_BF16_F16: copy_chunk<__bf16, _Float16>(seg.src, seg.dst, e0, e1); break;
        case P_F16_BF16: copy_chunk<_Float16, __bf16>(seg.src, seg.dst, e0, e1); break;
        default: break;  // rejected by rlx_copy_segments_plan
    }
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" int rlx_copy_segments_plan(rlx_copy_segment* table, int32_t n_segments, int64_t* total_chunks) {
    RLX_REQUIRE(n_segments >= 0 && (n_segments == 0 || table) && total_chunks, "rlx_copy_segments_plan: bad argument");
    int64_t chunks = 0;
    for (int32_t k = 0; k < n_segments; ++k) {
        rlx_copy_segment& s = table[k];
        RLX_REQUIRE(s.n >= 0, "rlx_copy_segments_plan: segment %d has a negative length", k);
        RLX_REQUIRE(path_of(s.src_dtype, s.dst_dtype) != P_BAD, "rlx_copy_segments_plan: segment %d: no conversion from dtype %d to %d",
                    k, s.src_dtype, s.dst_dtype);
        RLX_REQUIRE(s.n == 0 || (s.src && s.dst), "rlx_copy_segments_plan: segment %d: NULL pointer", k);
        RLX_REQUIRE(reinterpret_cast<uintptr_t>(s.src) % dtype_bytes(s.src_dtype) == 0 &&
                        reinterpret_cast<uintptr_t>(s.dst) % dtype_bytes(s.dst_dtype) == 0,
                    "rlx_copy_segments_plan: segment %d is not aligned to its element size", k);
        s.first_chunk = chunks;
        chunks += (s.n + RLX_COPY_CHUNK - 1) / RLX_COPY_CHUNK;
    }
    RLX_REQUIRE(chunks <= 0x7fffffffLL, "rlx_copy_segments_plan: %lld chunks exceed one grid dimension", (long long)chunks);
    *total_chunks = chunks;
    return RLX_OK;
}

extern "C" int rlx_copy_segments(const rlx_copy_segment* table_dev, int32_t n_segments, int64_t total_chunks,
                                 rlx_stream_t stream) {
    RLX_REQUIRE(n_segments >= 0 && total_chunks >= 0 && total_chunks <= 0x7fffffffLL, "rlx_copy_segments: bad argument");
    if (n_segments == 0 || total_chunks == 0) return RLX_OK;
    RLX_REQUIRE(table_dev, "rlx_copy_segments: NULL table");
    hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)total_chunks), dim3(CT), 0, static_cast<hipStream_t>(stream), table_dev,
                       n_segments);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
